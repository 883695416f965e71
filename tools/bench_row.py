#!/usr/bin/env python3
"""tools/bench_row.py FILE.json... -- one markdown table row per bench line (profiles/r06_bench_runs.md)."""
import json, sys
for f in sys.argv[1:]:
    d = json.load(open(f))
    r, m, s, t, g = d['roofline'], d['roofline_mfma'], d['secondary'], d['training'], d['dp_training']['gs']
    fr = lambda x: (x or {}).get
    rr, mr = r['second_ruler'], m['second_ruler']
    rib = g.get('rest_step_in_backward') or {}
    print('| `%s` | %.2f | %.3f | %.3f / %.3f / %.3f | %.3f / %.3f / %.3f | %.3f / %.3f (%.3f / %.3f) | %.3f / %.3f | %.3f / %.3f | %.3f / %.3f / %.3f | %.3f / %s / %.3f | %.1f |' % (
        f.split('/')[-1], d['value'], d['ms_per_step'], r['frac'], rr['in_frame_frac'], rr.get('rocprof_frac') or 0, m['frac'], mr['in_frame_frac'], mr.get('rocprof_frac') or 0,
        s['ms_fwd'], s['ms_fwd_bwd'], s['roofline']['frac_fwd'], s['roofline']['frac_fwd_bwd'], s['six_million']['ms_fwd'], s['six_million']['ms_fwd_bwd'],
        s['c3_1600x1060']['ms_fwd'], s['c3_1600x1060']['ms_fwd_bwd'], t['ms_per_iteration'], t['hip_graph']['ms_per_iteration'], t['fused']['ms_per_iteration'],
        g['ms_per_step'], ('%.3f' % rib['ms_per_step']) if rib else '—', g['hip_graph']['ms_per_step'], (d.get('secondary_trained') or {}).get('slab_order_auto', {}).get('mrays_per_s', float('nan'))))
