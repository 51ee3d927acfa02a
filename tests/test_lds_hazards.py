"""Build-time check of the compiled kernels (no GPU): no instruction may touch the destination registers of an LDS read that can
still be in flight (tools/check_lds_hazards.py has the analysis and the history: inline-asm reads the compiler does not track,
copied at loop exits while the data was on its way).  Runs on the objects `__graft_entry__.build()` / `make` left in
kalign_amd/csrc/build; skipped when there are none."""
import glob
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin/"


def test_no_use_of_lds_reads_in_flight():
    objs = sorted(glob.glob(os.path.join(ROOT, "kalign_amd", "csrc", "build", "ka_kernels_u*.o")))
    if not objs or not os.path.exists(LLVM + "llvm-objdump"):
        pytest.skip("no built kernel objects / no llvm-objdump")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_lds_hazards.py")] + objs, capture_output=True, text=True, timeout=900)
    tail = "\n".join(r.stdout.strip().split("\n")[-25:])
    assert r.returncode == 0, tail
    assert "suspicious uses in total: 0" in r.stdout, tail


def test_steady_octets_of_the_helper_wave_strips_carry_no_scratch_access():
    """tools/check_hot_loops.py on unit 0 (the 8-wave fast-mode task kernel): the steady-state octets of ka_wstrip -- what a pass
    spends its time in -- have no scratch_load / scratch_store, and the edge forms (which do spill) no more than a known bound."""
    obj = os.path.join(ROOT, "kalign_amd", "csrc", "build", "ka_kernels_u0.o")
    if not os.path.exists(obj) or not os.path.exists(LLVM + "llvm-objdump"):
        pytest.skip("no built kernel object / no llvm-objdump")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_hot_loops.py"), obj], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-1500:]
    assert "scratch-free:" in r.stdout


def test_steady_octets_of_the_lean_strips_carry_no_scratch_access():
    """the same on unit 10 (the throughput kernel, ka_lstrip.h): the strip is a real function per form with a register budget of 168
    -- its steady-state octets (5 / 20 / 23 residues, first and later strips, last row A or B) are scratch-free, the head and tail
    forms keep a handful of accesses per step"""
    obj = os.path.join(ROOT, "kalign_amd", "csrc", "build", "ka_kernels_u10.o")
    if not os.path.exists(obj) or not os.path.exists(LLVM + "llvm-objdump"):
        pytest.skip("no built kernel object / no llvm-objdump")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_hot_loops.py"), obj, "--steady", "6", "--max-scratch", "12"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-1500:]
    assert ".vgpr_count:     168" in r.stdout, r.stdout[:600]          # (three four-wave workgroups per CU)


def test_the_checker_sees_a_use_before_the_wait(tmp_path):
    """The analysis itself, on a ten-line kernel with the hazard built in (and the same kernel with the wait in place)."""
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc")
    src = r'''
#include <hip/hip_runtime.h>
typedef float float4v __attribute__((ext_vector_type(4)));
__global__ void k(float* out, int n)
{
    extern __shared__ float lds[];
    unsigned a = (unsigned)(unsigned long long)lds + threadIdx.x * 16;
    float4v q;
    float acc = 0;
    for (int i = 0; i < n; ++i) {
        asm volatile("ds_read_b128 %0, %1" : "=&v"(q) : "v"(a) : "memory");
#ifdef BAD
        acc += q.x;
#endif
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(q) : : "memory");
        acc += q.y;
    }
    out[threadIdx.x] = acc;
}
'''
    (tmp_path / "t.hip").write_text(src)
    for bad in (0, 1):
        obj = str(tmp_path / ("t%d.o" % bad))
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3"] + (["-DBAD"] if bad else []) + ["-c", "-o", obj, str(tmp_path / "t.hip")])
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_lds_hazards.py"), obj], capture_output=True, text=True, timeout=300)
        assert (r.returncode != 0) == bool(bad), r.stdout[-2000:]
