"""The register / scratch budget of the built kernels, read from the code objects inside librogue_gym_hip.so (no GPU needed: hipcc cross-compiles).

Two of the stepper's performance properties are decided by the compiler's register allocation and are easy to lose without noticing: the W <= 32 step
kernel must fit 256 registers (two waves per SIMD: every block of a 65 536-env launch resident from t = 0), and the observation
kernel must stay at 80 or fewer (six waves per SIMD; at 99 it lost 17 us, DESIGN_HISTORY.md 5.2).  No kernel may use scratch memory."""
import os
import re
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "rogue-gym_amd", "librogue_gym_hip.so")
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def kernel_metadata():
    tools = [os.path.join(LLVM, t) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-readelf")]
    if not os.path.exists(SO) or not all(os.path.exists(t) for t in tools):
        pytest.skip("library or LLVM tools not available")
    out = {}
    with tempfile.TemporaryDirectory() as d:
        fat = os.path.join(d, "fat.bin")
        subprocess.run([tools[0], "-O", "binary", "--only-section=.hip_fatbin", SO, fat], check=True)
        blob = open(fat, "rb").read()
        starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
        assert starts, "no offload bundle in the library"
        for i, s in enumerate(starts):  # one bundle per translation unit
            part = os.path.join(d, "b%d.bin" % i)
            open(part, "wb").write(blob[s:(starts[i + 1] if i + 1 < len(starts) else len(blob))])
            co = os.path.join(d, "b%d.co" % i)
            subprocess.run([tools[1], "--unbundle", "--type=o", "--input=" + part, "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], check=True,
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            notes = subprocess.run([tools[2], "--notes", co], check=True, capture_output=True, text=True).stdout
            for blk in notes.split("- .agpr_count:")[1:]:
                blk = ".agpr_count:" + blk
                f = {k: v for k, v in re.findall(r"\.(agpr_count|vgpr_count|sgpr_spill_count|vgpr_spill_count|private_segment_fixed_size|name):\s+(\S+)", blk)}
                if "name" in f:
                    out[f["name"]] = {k: int(v) for k, v in f.items() if k != "name"}
    return out


def test_register_and_scratch_budget_of_the_built_kernels():
    md = kernel_metadata()
    step32 = [k for k in md if "k_step_w32" in k]
    assert step32, sorted(md)
    for k in step32:
        m = md[k]
        assert m["vgpr_count"] <= 256 and m["agpr_count"] == 0, (k, m)   # (.vgpr_count is the unified total) two step waves per SIMD
        # (round 4, with the next-level structures: the allocator parks five entry-to-tail values in scratch memory -- stored once at the top of the wave,
        # read back once in its tail -- whatever is taken out of the turn's registers; measured against the unspilled build of the same day: 612.1 vs
        # 613.3 M steps/s, profiles/r04_experiments.txt.  A spill inside the turn's loops is what this guards against: it showed as 15+ registers.
        # Round 5, with the mirror update at the end of the turn: ten parked values.  A reload there is a LOAD behind every store of the turn: the update's
        # own addresses are computed in place for that reason, profiles/r05_experiments.txt.)
        # Round 6: the parked values were mostly ADDRESSES -- `field + e` for the SoA fields the tail stores, LDS columns at run-time offsets, the block's counter
        # row -- computed once at the top of the wave (GVN) and held for 30 000 instructions; the tail now addresses from an opaque copy of the env index, the
        # LDS columns sit at compile-time offsets, reciprocals of config constants come from RgConfig: 68 bytes of scratch -> 16, none of it reloaded behind
        # the turn's stores (the one parked pair is the address of the glyph table load, read back before the first store).
        assert m["vgpr_spill_count"] <= 5 and m["private_segment_fixed_size"] <= 20, (k, m)
    obs = [k for k in md if re.search(r"k_obsILi0ELb0ELb0E", k)]   # k_obs<gray, no config groups, not bound>: the kernel of the headline workload
    assert obs, sorted(md)
    for k in obs:
        # (rounds 2-5: 72 registers = seven waves per SIMD.  Round 6: with the register-to-store path of envs without a Redraw the kernel takes 79 -- six waves -- and
        # is FASTER for it: 28.9 us against 29.9 us for the 72-register form of the same path, same box; profiles/r06_experiments.txt)
        assert md[k]["vgpr_count"] <= 80 and md[k]["agpr_count"] == 0, (k, md[k])
    regen = [k for k in md if "k_regen" in k]
    assert regen, sorted(md)
    for k in regen:
        assert md[k]["vgpr_count"] <= 128, (k, md[k])              # two generator waves beside a step wave on a SIMD (248 + 2 x 96 <= 512)
    for k, m in md.items():
        if "huge" in k:  # the > 64-room instances (not a performance path): the compiler reserves a 68-byte frame for k_regen_huge that no instruction touches
            assert m["private_segment_fixed_size"] <= 128, ("scratch memory in", k, m)
            continue
        assert m["private_segment_fixed_size"] <= 20, ("scratch memory in", k, m)  # (a 20-byte frame no instruction touches, or k_step_w32's parked address)


def test_lds_dma_loads_name_their_lds_row_with_a_uniform_m0():
    """k_step parks a few per-env words in LDS with global_load_lds (no destination register, no wait): the LDS row of such a load is M0, wave-uniform by
    contract.  With those loads under divergent control flow the compiler once merged two of them into one instruction whose M0 came from v_readfirstlane of a
    PER-LANE value -- half the lanes wrote into the wrong row (profiles/r05_experiments.txt).  So: in the built code objects, no LDS-DMA load of a step kernel
    has a v_readfirstlane among the instructions that set up its M0."""
    objdump = os.path.join(LLVM, "llvm-objdump")
    tools = [os.path.join(LLVM, t) for t in ("llvm-objcopy", "clang-offload-bundler")]
    if not os.path.exists(SO) or not os.path.exists(objdump) or not all(os.path.exists(t) for t in tools):
        pytest.skip("library or LLVM tools not available")
    seen = 0
    with tempfile.TemporaryDirectory() as d:
        fat = os.path.join(d, "fat.bin")
        subprocess.run([tools[0], "-O", "binary", "--only-section=.hip_fatbin", SO, fat], check=True)
        blob = open(fat, "rb").read()
        starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
        for i, s0 in enumerate(starts):
            part, co = os.path.join(d, "b%d.bin" % i), os.path.join(d, "b%d.co" % i)
            open(part, "wb").write(blob[s0:(starts[i + 1] if i + 1 < len(starts) else len(blob))])
            subprocess.run([tools[1], "--unbundle", "--type=o", "--input=" + part, "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], check=True,
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            asm = subprocess.run([objdump, "-d", co], check=True, capture_output=True, text=True).stdout.splitlines()
            ins = [ln.split("//")[0].strip() for ln in asm if ln.startswith("\t")]
            for k, ln in enumerate(ins):
                if "global_load_lds" in ln:
                    seen += 1
                    # back to the s_mov that set M0 for this load: nothing on the way may be a readfirstlane
                    j = k - 1
                    while j >= 0 and k - j < 24 and not (ins[j].startswith("s_mov_b32 m0") or "m0," in ins[j]):
                        j -= 1
                    window = ins[max(0, j - 3):k]
                    assert not any("v_readfirstlane" in w for w in window), (i, k, window[-8:])
    assert seen >= 9, seen   # (the step kernels' parked loads are there at all)
