"""ISA gate on the built library: no packed-fp32 FMA anywhere.

Round-2 finding (tools/hazard_probe.hip, profiles/r02_hazard_probe.txt): on MI355X `v_pk_fma_f32` with an operand routed by
op_sel / op_sel_hi returns wrong lanes in ~2/3 of the launches while an MFMA kernel is co-resident on the same SIMDs, with a
full `s_waitcnt lgkmcnt(0)` and `s_nop 4` in front of it; the same FMAs as `v_pk_fma_f32` without op_sel, as `v_pk_mul_f32` +
`v_pk_add_f32`, or as scalar `v_fmac_f32` are always right.  hipcc forms the failing instruction from `float4 * scalar` code, so
the library is built with the packed-fp32 feature off (__graft_entry__.DEVICE_FLAGS) and this test keeps it that way."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def test_library_has_no_packed_fp32_fma(tmp_path):
    so = os.path.join(ROOT, "emotiongestures_amd", "libemogest_hip.so")
    if not os.path.exists(so):
        import __graft_entry__ as g
        g.build()
    if not os.path.exists(OBJDUMP):
        pytest.skip("llvm-objdump not available")
    local = tmp_path / "lib.so"
    shutil.copy(so, local)
    subprocess.run([OBJDUMP, "--offloading", str(local)], cwd=tmp_path, check=True, capture_output=True)
    objs = [f for f in os.listdir(tmp_path) if "hipv4-amdgcn" in f]
    assert objs, "no gfx950 code objects found in the library"
    n_mfma = 0
    for f in objs:
        dis = subprocess.run([OBJDUMP, "-d", "--mcpu=gfx950", str(tmp_path / f)], capture_output=True, text=True, check=True).stdout
        bad = [ln for ln in dis.splitlines() if "v_pk_fma_f32" in ln or "v_pk_mul_f32" in ln or "v_pk_add_f32" in ln]
        assert not bad, f"{f}: {len(bad)} packed-fp32 VALU instructions, e.g. {bad[0].strip()}"
        n_mfma += dis.count("v_mfma_f32")
    assert n_mfma > 1000          # the disassembly really covers the MFMA kernels
