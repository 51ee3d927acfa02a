"""CPU-side checks: the C-ABI library builds, loads and exports every symbol the header declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_header_symbols():
    import kalign_amd
    from kalign_amd import api
    if not os.path.exists(api.lib_path()):
        import __graft_entry__
        __graft_entry__.build()
    L = kalign_amd.load_library()
    hdr = open(os.path.join(ROOT, "include", "kalign_amd.h")).read()
    declared = set(re.findall(r"\b(ka_[a-z_]+)\s*\(", hdr))
    assert declared == set(api.EXPORTS)
    for name in declared:
        assert hasattr(L, name), name
    assert L.ka_abi_version() >= 1


def test_no_cpu_fallback_without_gpu():
    """Without a HIP device the product must fail loudly, not compute on the CPU."""
    import kalign_amd
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        pytest.skip("GPU present")
    with pytest.raises(kalign_amd.KalignAmdError):
        kalign_amd.Context(0)


def test_product_does_not_import_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "kalign_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "libkalign_oracle" not in src \
                    and "libkalign_ref" not in src, f


def test_header_is_plain_c_and_links(tmp_path):
    """The boundary is a C ABI: a C99 translation unit that includes the header, takes the address of every entry
    point and calls the ones that need no GPU must compile without warnings, link against the library and run."""
    import shutil
    import subprocess
    from kalign_amd import api
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    hdr = open(os.path.join(ROOT, "include", "kalign_amd.h")).read()
    names = sorted(set(re.findall(r"\b(ka_[a-z_]+)\s*\(", hdr)))
    src = tmp_path / "abi.c"
    src.write_text(
        '#include <stdio.h>\n#include "kalign_amd.h"\n'
        "static int dist(void* user, int n, const int* ia, const int* ib, int* out)\n"
        "{ int k; (void)user; for (k = 0; k < n; k++) out[k] = ia[k] > ib[k] ? ia[k] - ib[k] : ib[k] - ia[k]; return 0; }\n"
        "int main(void)\n{\n"
        "        void* fns[] = { " + ", ".join("(void*)" + n for n in names) + " };\n"
        "        int lens[3] = { 5, 4, 3 }, tasks[6];\n        float sd[3];\n        ka_task_rec rec;\n"
        "        (void)fns; (void)rec;\n"
        "        if (ka_abi_version() < 4) return 2;\n"
        "        if (ka_guide_tree_from(3, lens, dist, NULL, 1, NULL, tasks, sd)) { puts(ka_last_error()); return 3; }\n"
        "        if (tasks[5] != 4) return 4;                 /* two merges: nodes 3 and 4 */\n"
        '        if (!ka_guide_tree_from(1, lens, dist, NULL, 1, NULL, tasks, sd)) return 5;   /* one sequence: FAIL + message */\n'
        '        printf("%s\\n", ka_last_error());\n        return 0;\n}\n')
    exe = tmp_path / "abi"
    libdir = os.path.dirname(api.lib_path())
    cmd = ["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-Wno-pedantic", "-I", os.path.join(ROOT, "include"), str(src),
           "-o", str(exe), "-L", libdir, "-lkalign_amd", "-Wl,-rpath," + libdir, "-Wl,--allow-shlib-undefined"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(exe)], capture_output=True, text=True, env=dict(os.environ, LD_LIBRARY_PATH="/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", "")))
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert "bad arguments" in r.stdout


def test_dropin_exports_the_public_kalign_api():
    """oracle/_ref/dropin/libkalign.so.3 (the reference with the MI355X dispatcher underneath, `make -C oracle dropin`)
    loads without a GPU and exports every function of lib/include/kalign/kalign.h:36-109 -- the API north_star says
    must survive -- with SONAME libkalign.so.3; the four replaced seams come from the glue, the reference's own
    definitions are still there under their kalign_ref_ names."""
    import ctypes as C
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = os.path.join(root, "oracle", "_ref", "dropin", "libkalign.so.3")
    if not os.path.exists(so):
        import pytest
        pytest.skip("drop-in library not built (needs the reference sources: make -C oracle dropin)")
    L = C.CDLL(so)
    for name in ("kalign_read_input", "kalign_write_msa", "kalign", "kalign_run", "kalign_run_seeded", "kalign_run_dist_scale",
                 "kalign_run_realign", "kalign_post_realign", "kalign_ensemble", "kalign_consensus_from_poar",
                 "kalign_free_msa", "reformat_settings_msa", "kalign_check_msa", "kalign_msa_compare",
                 "kalign_msa_compare_detailed", "kalign_msa_compare_with_mask", "kalign_arr_to_msa", "kalign_msa_to_arr"):
        assert hasattr(L, name), name
    for name in ("create_msa_tree", "anchor_consistency_build", "build_tree_kmeans", "finalise_alignment"):
        assert hasattr(L, name) and hasattr(L, "kalign_ref_" + name), name
    dyn = subprocess.run(["readelf", "-d", so], stdout=subprocess.PIPE).stdout.decode()
    assert "libkalign.so.3" in dyn and "libkalign_amd.so" in dyn
