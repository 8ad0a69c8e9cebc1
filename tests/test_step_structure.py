"""ONE launch structure per (batch regime, datatype, batch_norm, data-parallel form, option fused_launches).  The table is
restated here and compared with what the library reports for a handle of every combination (sdqn_net_step_structure) — a combination that
starts taking another path fails this test before it changes a number."""
import itertools

import pytest

from util import make_args

pytestmark = pytest.mark.gpu


def expected(B, datatype, bn, dp, fused):
    if dp == "overlap" and fused:
        return "dp_overlap", "dp_overlap"
    upd = {"none": "single", "serial": "dp_serial", "overlap": "dp_serial", "grad_only": "grad_only"}[dp]
    if datatype == "float16" and B >= 128 and not bn and fused:
        return "h16_block_tile", upd
    return ("fused" if fused else "unfused"), upd


@pytest.fixture(scope="module")
def sd():
    import simple_dqn_amd
    return simple_dqn_amd


def test_every_reachable_combination_takes_the_tabulated_structure(sd):
    from simple_dqn_amd.deepqnetwork import dp_unique_id
    seen = set()
    for B, datatype, bn, dp, fused in itertools.product((32, 256), ("float32", "float16"), (False, True), ("none", "serial", "overlap", "grad_only"), (1, 0)):
        if bn and datatype == "float16":
            continue                                             # refused at creation (--batch_norm with float16)
        net = sd.DeepQNetwork(3, make_args(batch_size=B, datatype=datatype, batch_norm=bn))
        net.set_option("fused_launches", fused)
        if dp == "grad_only":
            net.set_option("grad_only", 1)
        if dp in ("serial", "overlap"):
            net.set_option("dp_overlap", 1 if dp == "overlap" else 0)
            net.dp_init(dp_unique_id(), 0, 1)
        got = net.step_structure()
        assert got == expected(B, datatype, bn, dp, fused), (B, datatype, bn, dp, fused, got)
        seen.add(got)
        if dp in ("serial", "overlap"):
            net.dp_shutdown()
        del net
    assert seen == {("fused", "single"), ("fused", "dp_serial"), ("fused", "grad_only"), ("unfused", "single"), ("unfused", "dp_serial"),
                    ("unfused", "grad_only"), ("h16_block_tile", "single"), ("h16_block_tile", "dp_serial"), ("h16_block_tile", "grad_only"),
                    ("dp_overlap", "dp_overlap")}
    g = sd.DeepQNetwork(3, make_args(batch_size=8, datatype="float64"))
    assert g.step_structure() == ("generic", "single")
