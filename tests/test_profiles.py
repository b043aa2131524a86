"""Hygiene of the committed measurements: the counter-based roofline figure bench.py reports comes from
profiles/pmc_latest.json, which is only valid for the kernel sources it was collected on.  bench.py flags a
mismatch at run time (roofline.traffic_stale); this test flags it at commit time."""
import json
import os

from simple_mip_solver_amd import _ffi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_source_hash_is_a_sha256_of_the_kernel_sources():
    h = _ffi.source_hash()
    assert len(h) == 64 and int(h, 16) >= 0 and h == _ffi.source_hash()


def test_pmc_latest_was_collected_on_the_committed_kernel_sources():
    pmc = json.load(open(os.path.join(ROOT, 'profiles', 'pmc_latest.json')))
    assert pmc.get('hbm_bytes_per_launch', 0) > 0 and 'FETCH_SIZE' in pmc['command'] and 'WRITE_SIZE' in pmc['command']
    assert pmc.get('csrc_sha256') == _ffi.source_hash(), \
        'simple_mip_solver_amd/csrc changed since profiles/pmc_latest.json was collected: re-run scripts/profile_bench.sh ' \
        'on the GPU box and copy pmc_latest.json (bench.py reports roofline.traffic_stale until then)'
