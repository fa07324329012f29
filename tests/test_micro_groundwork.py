"""CPU-side checks of the round-4 groundwork under tools/micro/ (DESIGN.md 9.1-9.2): NOT product code -- these tests only make the
claims of that section reproducible: the strip kernel's index arithmetic, the DMA / barrier schedules of the eight-wave tiles in
the barrier-interval model (and that the model flags broken schedules), and that the kernels compile for gfx950 without scratch
(a scratch reload is a VMEM operation and would corrupt their counted vmcnt waits)."""
import os
import re
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MICRO = os.path.join(ROOT, "tools", "micro")


def test_strip_kernel_index_arithmetic():
    r = subprocess.run([sys.executable, os.path.join(MICRO, "strip_index_check.py")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "conflict-free for every start row: True" in r.stdout and "FAILED" not in r.stdout


def test_tile_schedules_in_the_barrier_interval_model():
    sys.path.insert(0, MICRO)
    try:
        import schedule_check as S
    finally:
        sys.path.pop(0)
    for sc in [S.gemm256(2, 5), S.gemm256(36, 5), S.gemm256(2, 7), S.gemm256(36, 7), S.gemm256x128(3), S.gemm256x128(36), S.strip(6), S.strip(36)]:
        assert sc.check() == [], (sc.name, sc.check()[:3])
    # the model has teeth: a looser wait in the strip schedule, and a six-slot ring for the 256 x 128 tile, are violations
    s = S.strip(18)
    for P in s.phases:
        P["wait"] = 10
    assert any(e.startswith("RAW") for e in s.check())
    s = S.gemm256x128(9)
    fix = lambda l: [(it, (r % 6) if isinstance(r, int) else r) for it, r in l]
    s.pro_issues = fix(s.pro_issues)
    for P in s.phases:
        P["issues"] = fix(P["issues"])
    assert any(e.startswith("WAR") for e in s.check())
    # issuing every half-tile one phase later than ring depth 5 allows breaks read-after-write
    s = S.gemm256(8, 5)
    for ph in range(len(s.phases) - 1, 0, -1):
        s.phases[ph]["issues"] = s.phases[ph - 1]["issues"]
    s.phases[0]["issues"] = []
    assert any(e.startswith("RAW") for e in s.check())


@pytest.mark.parametrize("src,flags", [("gemm256_stagger.hip", ["-mllvm", "-amdgpu-mfma-vgpr-form"]), ("gemm256_strip.hip", [])])
def test_groundwork_kernels_compile_without_scratch(src, flags, tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    out = tmp_path / "k.s"
    r = subprocess.run([hipcc, "-O3", "--offload-arch=gfx950", *flags, "--cuda-device-only", "-S", os.path.join(MICRO, src), "-o", str(out)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    asm = out.read_text()
    kernels = re.findall(r"^(_Z\d+gemm256\w*):", asm, flags=re.M)
    scratch = re.findall(r"; ScratchSize: (\d+)", asm)
    assert len(kernels) >= 3 and len(scratch) >= len(kernels)
    assert all(int(x) == 0 for x in scratch), scratch
    if src == "gemm256_strip.hip":
        body = asm[asm.index("_Z13gemm256_stripILb1ELb0EEv3Prm:"):]
        body = body[:body.index("s_endpgm")]
        waits = re.findall(r"s_waitcnt vmcnt\((\d+)\)", body)
        # 24 phases + prologue with the schedule's constants, one final drain, nothing inserted by the compiler
        assert sorted(set(waits)) == ["0", "10", "8"] and waits.count("0") == 1 and waits.count("8") == 13 and waits.count("10") == 12, waits
        assert body.count("s_barrier") == 52 and body.count("v_mfma_f32_32x32x16_bf16") == 192


def test_library_gemm256_schedule_and_isa(tmp_path):
    """csrc/gemm256.hip (the product's eight-wave tile) runs the D = 5 schedule of gemm256_stagger.hip: the barrier-interval model
    is clean for every K-tile count the launcher can produce (even, odd, the two-tile minimum of a split), and the gfx950 build
    has no scratch in either instantiation (a scratch reload is a VMEM operation: it would corrupt the counted vmcnt waits),
    only the schedule's wait constants, and descriptor-addressed LDS-DMA for every operand load."""
    sys.path.insert(0, MICRO)
    try:
        import schedule_check as S
    finally:
        sys.path.pop(0)
    for kt in (2, 3, 4, 5, 9, 18, 19, 27, 36, 38, 72, 76, 152):
        assert S.gemm256(kt, 5).check() == [], kt
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    out = tmp_path / "g.s"
    src = os.path.join(ROOT, "ddpm_image_restoration_amd", "csrc", "gemm256.hip")
    r = subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-gpu-rdc", "-mllvm", "-amdgpu-mfma-vgpr-form", "--cuda-device-only",
                        "-S", src, "-o", str(out)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    asm = out.read_text()
    names = re.findall(r"^(_ZN6ddpmir\S*gemm256_kernelILb[01]E\S*):", asm, flags=re.M)
    assert len(names) == 2, names
    for nm in names:
        body = asm[asm.index(nm + ":"):]
        body = body[:body.index("s_endpgm")]
        meta = asm[asm.index(".name:           " + nm):]
        meta = meta[:meta.index(".wavefront_size")] if ".wavefront_size" in meta else meta[:3000]
        assert re.search(r"\.private_segment_fixed_size:\s+0\b", meta) and re.search(r"\.vgpr_spill_count:\s+0\b", meta), nm
        assert "scratch_" not in body and "buffer_load_dwordx4" in body and "global_load_lds" not in body
        # counted waits of the K loop: vmcnt(4) steady state, vmcnt(2) / vmcnt(0) in the tail; never anything else between barriers
        loop = body[:body.rindex("s_barrier")]
        assert set(re.findall(r"s_waitcnt vmcnt\((\d+)\)", loop)) <= {"0", "2", "4"}, set(re.findall(r"s_waitcnt vmcnt\((\d+)\)", loop))
        assert loop.count("s_waitcnt vmcnt(4)") >= 9      # prologue + the eight phases of the unrolled pair of K tiles
        # bf16 kernel: 8 + 4 phases x 8 MFMAs; e4m3 kernel: 8 + 4 e4m3 phases x 4 MFMAs (64-deep) + 8 bf16 phases of the second segment
        if "ILb1E" in nm: assert body.count("32x32x64_f8f6f4") == 48 and body.count("v_mfma_f32_32x32x16_bf16") == 64
        else: assert body.count("v_mfma_f32_32x32x16_bf16") == 96


def test_frequency_kernel_isa(tmp_path):
    """csrc/freq_fused.hip is bound by VALU issue, so its instruction count is a tracked quantity: the C = 64
    kernel stays near the 2074 instructions round 4 brought it to (from 3744), nothing spills in the instantiations the models
    launch at full resolution, and no v_pk_fma_f32 comes back (the packed DCT was wrong with several workgroups per CU)."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    out = tmp_path / "f.s"
    src = os.path.join(ROOT, "ddpm_image_restoration_amd", "csrc", "freq_fused.hip")
    r = subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-gpu-rdc", "-mllvm", "-amdgpu-mfma-vgpr-form", "--cuda-device-only",
                        "-S", src, "-o", str(out)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    asm = out.read_text()
    names = re.findall(r"^(_ZN6ddpmir\S*freq_gate_fused_kILi\dELi\d+ELb[01]E\S*):", asm, flags=re.M)
    assert len(names) == 8, names
    count = {}
    for nm in names:
        body = asm[asm.index(nm + ":"):]
        body = body[:body.index("s_endpgm")]
        key = re.search(r"kILi(\d)ELi(\d+)ELb([01])E", nm).groups()
        count[key] = sum(1 for l in body.split("\n") if l.strip() and not l.strip().startswith((".", ";", "_Z")))
        assert "v_pk_fma_f32" not in body, nm
        meta = asm[asm.index(".name:           " + nm):][:3000]
        if key != ("8", "128", "1"):      # (the 128-channel kernel with the projection inside keeps a handful of spilled registers)
            assert re.search(r"\.vgpr_spill_count:\s+0\b", meta), nm
        assert int(re.search(r"\.vgpr_count:\s+(\d+)", meta).group(1)) <= 168, nm        # three workgroups per CU
    assert count[("8", "64", "0")] < 2300 and count[("8", "64", "1")] < 2650 and count[("4", "64", "0")] < 1900, count
    # no register is read or overwritten while the load that writes it may still be in flight (tools/micro/waitcnt_check.py);
    # the checker itself has teeth: weakening every lgkmcnt(0) by one is noticed
    sys.path.insert(0, MICRO)
    try:
        import waitcnt_check as Wc
    finally:
        sys.path.pop(0)
    seen = 0
    for nm, body in Wc.kernels(asm):
        if "freq_gate_fused_k" not in nm:
            continue
        seen += 1
        assert Wc.check(body, nm, verbose=False) == [], nm
        if "ILi8ELi64ELb0E" in nm:
            assert len(Wc.check([l.replace("lgkmcnt(0)", "lgkmcnt(1)") for l in body], nm, verbose=False)) > 100
    assert seen == 8
