"""CPU-side guard of the hand-placed matrix-pipe hazards in csrc/d3k_conv.hpp: the kernel's MFMAs are
inline asm (their A operand must come from AGPRs), so hipcc inserts none of the wait states between an MFMA and a vector instruction
that reads its accumulator. Two wrong-result bugs of that kind were only visible to the bit-exact GPU test; this test disassembles
what the compiler emits and measures the distances, so a toolchain or flag change that re-breaks them fails HERE, without a GPU."""
import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
import isa_hazards  # noqa: E402

SRC = os.path.join(ROOT, "pytorchcv_amd", "csrc", "d3k_16bit.hip")
# XDL write -> vector read on gfx950: passes + 3 wait states; v_mfma_f32_16x16x32_{f16,bf16} is budgeted as an 8-pass instruction
REQUIRED = 11

pytestmark = pytest.mark.skipif(not os.path.exists(isa_hazards.HIPCC), reason="hipcc not installed")


@pytest.fixture(scope="module")
def shipped(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("isa") / "d3k.s")
    return isa_hazards.kernels(isa_hazards.device_asm(SRC, out))


def test_d3k_kernels_are_found_and_use_inline_asm_mfmas(shipped):
    assert len(shipped) == 2, sorted(shipped)                      # bf16 and fp16
    for name, insts in shipped.items():
        n_asm = sum(1 for op, _, in_asm in insts if in_asm and isa_hazards.is_mfma(op))
        assert n_asm >= 2 * 252, (name, n_asm)                     # 252 steps x 2 channel halves per tile body


def test_d3k_accumulator_reads_keep_their_distance_from_the_asm_mfmas(shipped):
    for name, insts in shipped.items():
        d = isa_hazards.asm_mfma_distances(insts)
        assert d, name
        worst = min(d, key=lambda t: t[2])
        assert worst[2] >= REQUIRED, "{}: {} reads an accumulator {} wait states behind its MFMA (instruction {} -> {}), need {}".format(
            name, worst[3], worst[2], worst[0], worst[1], REQUIRED)


def test_d3k_weights_stay_in_agprs_and_nothing_spills(shipped, tmp_path):
    asm = isa_hazards.device_asm(SRC, str(tmp_path / "d3k.s"))
    assert "v_accvgpr_read" not in asm and "v_accvgpr_write" not in asm      # no AGPR <-> VGPR copies of the register-resident weights
    assert "scratch_" not in asm
    for line in asm.split("\n"):
        if ".vgpr_spill_count:" in line or ".private_segment_fixed_size:" in line:
            assert line.split(":")[1].strip() == "0", line


def test_the_checker_notices_missing_pads(tmp_path):
    """The same source without the hand-placed wait states (-DD3K_DROP_HAZARD_PADS) must FAIL the distance check: the test has teeth."""
    asm = isa_hazards.device_asm(SRC, str(tmp_path / "d3k_nopad.s"), extra=["-DD3K_DROP_HAZARD_PADS"])
    worst = min(min(t[2] for t in isa_hazards.asm_mfma_distances(insts)) for insts in isa_hazards.kernels(asm).values())
    assert worst < REQUIRED, worst
