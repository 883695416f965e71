#!/usr/bin/env python3
"""tools/make_profile_summary.py -- turns gpurun_out/profiles_raw (tools/collect_profiles.sh) into the committed summaries:
profiles/r02_bench_kernel_stats.csv, r02_gs_kernel_stats.csv (1 M Gaussians), r02_gs6m_kernel_stats.csv, r02_train_kernel_stats.csv,
profiles/r02_pmc_summary.md, profiles/pmc_summary.json (read by bench.py for roofline.traffic and the per-kernel 3DGS entries)."""
import collections, csv, glob, json, re, shutil, sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
RAW = ROOT / 'gpurun_out' / 'profiles_raw'
OUT = ROOT / 'profiles'
OUT.mkdir(exist_ok=True)
ROUND = sys.argv[1] if len(sys.argv) > 1 else 'r03'   # file prefix of the committed summaries


def short(name):
    m = re.match(r'_ZN12_GLOBAL__N_1(\d+)', name)  # rocprofv3 leaves some template instances mangled
    if m:
        n = int(m.group(1))
        base, rest = name[m.end():m.end() + n], name[m.end() + n:]
        t = re.match(r'I((?:L[ib]\d+E)+)E', rest)
        if t:
            base += '<' + ', '.join(re.findall(r'L[ib](\d+)E', t.group(1))) + '>'
        return base
    name = re.sub(r'\(anonymous namespace\)::', '', name).replace('void ', '')
    return name.split('(')[0]


import os
stats = sorted(glob.glob(str(RAW / 'bench_stats' / '*' / '*kernel_stats.csv')), key=os.path.getmtime)  # newest collection last
if stats:
    shutil.copy(stats[-1], OUT / f'{ROUND}_bench_kernel_stats.csv')
    # per-kernel mean durations of the bench command as JSON (bench.py's second ruler: roofline.second_ruler.rocprof_ms), gated on the source digests below
    import hashlib as _hl
    durations = {}
    for r in csv.DictReader(open(stats[-1])):
        durations[short(r['Name'])] = {'calls': int(r['Calls']), 'avg_us': round(float(r['AverageNs']) / 1e3, 3), 'total_us': round(float(r['TotalDurationNs']) / 1e3, 1)}
    durations['_meta'] = {'round': ROUND, 'command': 'python bench.py --steps 20 --warmup 5',
                          'csrc_sha': {p.name: _hl.sha256(p.read_bytes()).hexdigest()[:16] for p in sorted((ROOT / 'nerficg_amd' / 'csrc').glob('*.h*'))}}
    (OUT / 'kernel_durations.json').write_text(json.dumps(durations, indent=1, sort_keys=True))
for tag, name in (('gs_stats', f'{ROUND}_gs_kernel_stats.csv'), ('gs6_stats', f'{ROUND}_gs6m_kernel_stats.csv')):
    extra = sorted(glob.glob(str(RAW / tag / '*' / '*kernel_stats.csv')), key=os.path.getmtime)
    if extra:
        shutil.copy(extra[-1], OUT / name)
tstats = sorted(glob.glob(str(RAW / 'train_stats' / '*' / '*kernel_stats.csv')), key=os.path.getmtime)
if tstats:
    shutil.copy(tstats[-1], OUT / f'{ROUND}_train_kernel_stats.csv')
fstats = sorted(glob.glob(str(RAW / 'fused_stats' / '*' / '*kernel_stats.csv')), key=os.path.getmtime)
if fstats:
    shutil.copy(fstats[-1], OUT / f'{ROUND}_fused_train_kernel_stats.csv')
# one accumulator per collection run: pmc_* = tools/bench_query.py (InstantNGP image pipeline), pmcgs_* = tools/bench_gs.py (3DGS frame),
# pmctr_* = tools/bench_train.py (InstantNGP training iteration).  A kernel that appears in several of them (k_grid_encode runs in the image
# pipeline with 8 Mi-slot launches and in training with 264 K samples) is reported from the run that is about it.
by_run = {tag: collections.defaultdict(lambda: collections.defaultdict(list)) for tag in ('pmc_', 'pmcgs_', 'pmctr_', 'pmcfu_')}
for f in glob.glob(str(RAW / 'pmc*' / '*' / '*counter_collection.csv')):
    tag = 'pmcgs_' if '/pmcgs_' in f else ('pmctr_' if '/pmctr_' in f else ('pmcfu_' if '/pmcfu_' in f else 'pmc_'))
    for r in csv.DictReader(open(f)):
        by_run[tag][short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
home = {'k_grid_encode<1': 'pmc_', 'k_ngp_mlp': 'pmc_', 'k_render_count': 'pmc_', 'k_render_write': 'pmc_', 'k_composite_image': 'pmc_', 'k_ray_sh': 'pmc_',
        'k_grid_encode<0': 'pmctr_', 'k_nwie': 'pmctr_', 'k_grid_bwd': 'pmctr_', 'k_gb_': 'pmctr_', 'k_march': 'pmctr_', 'k_composite_train': 'pmctr_', 'k_adam': 'pmctr_', 'k_train_': 'pmcfu_', 'k_amp_': 'pmcfu_'}
acc = collections.defaultdict(lambda: collections.defaultdict(list))
names = set().union(*[set(v) for v in by_run.values()])
for k in names:
    pref = next((t for p, t in home.items() if k.startswith(p)), 'pmcgs_')
    src = by_run[pref] if k in by_run[pref] else next(v for v in by_run.values() if k in v)
    for n, vals in src[k].items():
        acc[k][n] = vals
keep = ('k_grid_encode', 'k_ngp_mlp', 'k_render', 'k_composite_image', 'k_preprocess', 'k_span_', 'k_item_', 'k_depth_keys', 'k_radix_', 'k_scan_tiles', 'k_march_wave',
        'k_grid_bwd', 'k_nwie_', 'k_composite_train', 'k_gb_', 'k_train_', 'k_amp_', 'k_adam')
lines = [f'# rocprofv3 --pmc summary (MI355X, {ROUND})', '',
         'Collected by `tools/collect_profiles.sh` (one `--pmc` group per run, `--kernel-trace` only), averaged per kernel over all launches of',
         '`tools/bench_query.py` (InstantNGP 800x800 image pipeline), `tools/bench_gs.py` (3DGS, 1 M Gaussians, 1297x840) and `tools/bench_train.py`',
         '(InstantNGP training iteration, 2200 rays / 264 K samples).',
         'FETCH_SIZE / WRITE_SIZE are in KiB as reported; per MI355X_MICROARCH.md FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950',
         '(16-byte gathers: the same 2x -- every L2 miss is ONE 128-byte fabric request tallied at 64 B, tools/micro/gather_fetch_calib.hip,',
         'profiles/r03_gather_calibration.md) -- both the raw value and the 2x-corrected read bytes are listed.', '']
summary = {}
for k in sorted(acc):
    if not any(k.startswith(p) for p in keep):
        continue
    c = {n: sum(v) / len(v) for n, v in acc[k].items()}
    lines.append(f'## {k}')
    lines.append('')
    lines.append('| counter | mean per launch | launches |')
    lines.append('|---|---|---|')
    for n in sorted(c):
        lines.append(f'| {n} | {c[n]:.1f} | {len(acc[k][n])} |')
    fetch, write = c.get('FETCH_SIZE'), c.get('WRITE_SIZE')
    derived = {}
    if fetch is not None and write is not None:
        derived['hbm_bytes_per_launch_raw'] = int((fetch + write) * 1024)
        derived['hbm_bytes_per_launch'] = int((2 * fetch + write) * 1024)
    if 'TCP_TOTAL_CACHE_ACCESSES_sum' in c and 'GRBM_GUI_ACTIVE' in c:
        cyc = c['GRBM_GUI_ACTIVE'] / 8.0  # sum over 8 XCDs
        derived['kernel_cycles'] = int(cyc)
        derived['tcp_accesses_per_clk_per_cu'] = round(c['TCP_TOTAL_CACHE_ACCESSES_sum'] / 256.0 / cyc, 3)
        if 'TCP_TCC_READ_REQ_sum' in c:
            derived['l1_hit_rate'] = round(1 - c['TCP_TCC_READ_REQ_sum'] / c['TCP_TOTAL_CACHE_ACCESSES_sum'], 3)
    if 'TCC_HIT_sum' in c and 'TCC_MISS_sum' in c and c['TCC_HIT_sum'] + c['TCC_MISS_sum'] > 0:
        derived['l2_hit_rate'] = round(c['TCC_HIT_sum'] / (c['TCC_HIT_sum'] + c['TCC_MISS_sum']), 3)
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in c and 'GRBM_GUI_ACTIVE' in c and c['GRBM_GUI_ACTIVE'] > 0:
        # busy cycles are summed over the 1024 SIMDs (checked: = MFMA count x 32 cycles); kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs
        derived['mfma_pipe_busy_frac'] = round(c['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024.0 * c['GRBM_GUI_ACTIVE'] / 8.0), 3)
    if 'SQ_INSTS_VALU_MFMA_MOPS_F16' in c and 'GRBM_GUI_ACTIVE' in c and c['GRBM_GUI_ACTIVE'] > 0:
        # one MOP = 512 FLOP; clock taken as 2.4 GHz
        derived['mfma_tflops_issued'] = round(c['SQ_INSTS_VALU_MFMA_MOPS_F16'] * 512 / (c['GRBM_GUI_ACTIVE'] / 8.0 / 2.4e9) / 1e12, 1)
    if derived:
        lines.append('')
        lines.append('derived: ' + ', '.join(f'{a} = {b}' for a, b in derived.items()))
    lines.append('')
    key = k
    if k.startswith('k_grid_encode') or k.startswith('k_ngp_mlp'):
        key = re.sub(r'<1(, \d+)?>', '<SRC_TILED>', re.sub(r'<0(, \d+)?>', '<SRC_ARRAYS>', k))
    summary[key] = {**{n: round(v, 1) for n, v in c.items()}, **derived}
(OUT / f'{ROUND}_pmc_summary.md').write_text('\n'.join(lines))
import hashlib
# provenance: bench.py quotes a counter entry only while the kernel's source file is the one these counters were collected on
summary['_meta'] = {'round': ROUND, 'csrc_sha': {p.name: hashlib.sha256(p.read_bytes()).hexdigest()[:16] for p in sorted((ROOT / 'nerficg_amd' / 'csrc').glob('*.h*'))}}
(OUT / 'pmc_summary.json').write_text(json.dumps(summary, indent=1, sort_keys=True))
print('\n'.join(lines[:8]))
for k, v in summary.items():
    if k.startswith('_'):
        continue
    print(k, {a: v[a] for a in ('hbm_bytes_per_launch', 'tcp_accesses_per_clk_per_cu', 'l1_hit_rate', 'l2_hit_rate', 'mfma_pipe_busy_frac', 'mfma_tflops_issued') if a in v})
