"""CPU: the committed counter / duration summaries belong to the committed kernel sources.

bench.py quotes `roofline.traffic`, `limiter`, `roofline_valu` and the rocprofv3 durations from profiles/pmc_summary.json / profiles/kernel_durations.json only for kernels
whose source file has the digest recorded at collection time (bench.csrc_digests) -- an honest gate, but one that silently empties those fields in the DRIVER's line when a
source file is touched after the collection (rounds 4 and 5 both ended that way: a late commit edited gs_raster.hip behind the counters).  This test makes the tree say so:
it fails as long as any digest differs, i.e. until tools/collect_profiles.sh + tools/make_profile_summary.py have been re-run on the final sources."""
import json
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


@pytest.mark.parametrize('name', ['pmc_summary.json', 'kernel_durations.json'])
def test_profile_summaries_were_collected_on_the_sources_in_the_tree(name):
    import bench
    f = ROOT / 'profiles' / name
    assert f.exists(), f'{f} is missing: run tools/collect_profiles.sh on the GPU box and tools/make_profile_summary.py here'
    recorded = json.loads(f.read_text()).get('_meta', {}).get('csrc_sha', {})
    now = bench.csrc_digests()
    stale = sorted(k for k in now if recorded.get(k) != now[k])
    assert not stale, (f'profiles/{name} was collected on other sources than the tree holds: {stale} changed since -- bench.py would not quote their kernels '
                       '(traffic / roofline_valu / rocprof durations null in the driver\'s line).  Re-collect: tools/collect_profiles.sh, tools/make_profile_summary.py')


def test_the_bench_gate_uses_the_same_digests():
    """pmc_entry refuses an entry exactly when the digest differs (the rule the test above enforces for the whole tree)."""
    import bench
    d = bench.csrc_digests()
    fake = {'_meta': {'round': 'x', 'csrc_sha': dict(d)}, 'k_render': {'hbm_bytes_per_launch': 1}}
    assert bench.pmc_entry(fake, 'k_render', 'gs_raster.hip')[0] == {'hbm_bytes_per_launch': 1}
    fake['_meta']['csrc_sha']['gs_raster.hip'] = 'other'
    entry, why = bench.pmc_entry(fake, 'k_render', 'gs_raster.hip')
    assert entry == {} and 'predates' in why
