"""Static properties of the product's gfx950 code that the measured numbers rest on, checked WITHOUT a GPU (tools/isa_census.py compiles a
translation unit to assembly and reads the compiler's own per-kernel resource comments).  A compiler or source change that silently spills,
halves a launch's residency or brings back the self-draining prefetch ring fails here, not in a later round's profile."""
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_census  # noqa: E402

pytestmark = pytest.mark.skipif(not os.path.exists(isa_census.HIPCC), reason="hipcc not installed")


@pytest.fixture(scope="module")
def rows():
    with ThreadPoolExecutor(2) as ex:                       # two hipcc processes side by side: ~1 minute
        r3, bt = ex.map(isa_census.census_rows, ["sdqn_kernels_r3.hip", "sdqn_kernels_bt.hip"])
    return {"r3": r3, "bt": bt}


def _find(rows, *parts):
    hit = [r for r in rows if all(p in r["name"] for p in parts)]
    assert len(hit) == 1, (parts, [r["name"][:90] for r in hit])
    return hit[0]


def test_no_kernel_of_the_step_uses_scratch(rows):
    bad = [(r["name"][:80], r["scratch"]) for tu in rows.values() for r in tu if r["scratch"]]
    assert not bad, bad


def test_residency_of_the_b32_launches(rows):
    """B = 32: every workgroup of a launch is resident in ONE round: two 512-thread workgroups of bwd3 / bwd2 per CU need
    <= 128 VGPRs and <= 80 KB of LDS each, the 1024-thread forward tiles two per CU need <= 64 VGPRs."""
    r3 = rows["r3"]
    bwd3 = _find(r3, "gemm_multi_kernelILi512E", "Conv3DgradWT", "Fc4WgradWTELi1E")
    bwd2 = _find(r3, "gemm_multi_kernelILi512ENS_9NoProblemELi2ENS_12Conv2DgradWTELi8E")
    for k in (bwd3, bwd2):
        assert k["vgpr"] + k["agpr"] <= 128 and k["lds"] <= 80 * 1024, (k["name"][:60], k["vgpr"], k["lds"])
    conv2 = _find(r3, "gemm_kernelINS_10Conv2FwdWTELi16E")
    assert conv2["vgpr"] + conv2["agpr"] <= 64 and conv2["lds"] <= 80 * 1024, (conv2["vgpr"], conv2["lds"])
    conv1 = _find(r3, "conv1_bf16_kernelILb1E")
    assert conv1["lds"] <= 80 * 1024                        # two workgroups per CU beside the 50 KB of weight planes


def test_block_tile_rings_do_not_drain_their_own_loads(rows):
    """B >= 128 (round 4): the weight-gradient problems have run-time chunk counts; with the GUARDED ring load hipcc waited with vmcnt(3 .. 0)
    in front of every LDS store — the loads just issued were drained, one memory round trip per chunk.  The product launches
    use unconditional clamped loads: their waits leave the newest loads in flight (vmcnt(4 ..) dominate), the guarded forms kept in the menu
    for A/B runs show the old picture."""
    bt = rows["bt"]
    # (the trailing template arguments of BtCfg are X, CPI, UNC, [PIPE]: "...Li0ELi1ELi1EEE" = unconditional ring loads)
    prod = [r for r in bt if "bt_multi_kernel" in r["name"] and "Conv3DgradWTELi64ELi64ELi2ELi2ELi2E" in r["name"] and "Conv3WgradWTELi64ELi64ELi2ELi2ELi2ELi0ELi1ELi1E" in r["name"]]
    old = [r for r in bt if "bt_multi_kernel" in r["name"] and "Conv3DgradWTELi64ELi64ELi2ELi2ELi2E" in r["name"] and "Conv3WgradWTELi64ELi64ELi2ELi2ELi2ELi0ELi1ELi0E" in r["name"]]
    assert prod and old
    drained = lambda r: sum(c for n, c in r["vmcnt"].items() if n <= 3)
    ahead = lambda r: sum(c for n, c in r["vmcnt"].items() if 4 <= n <= 7)
    assert min(ahead(r) for r in prod) > max(ahead(r) for r in old)
    assert max(r["vmcnt"][0] for r in prod) < min(r["vmcnt"][0] for r in old)
    assert all(drained(r) < ahead(r) for r in prod)
    # every block-tile launch of the default step keeps >= 3 workgroups of 256 threads per CU in registers
    for r in prod:
        assert r["vgpr"] + r["agpr"] <= 160, (r["name"][:60], r["vgpr"], r["agpr"])
