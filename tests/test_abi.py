"""The C-ABI libraries load and export every symbol include/*.h declares (no compute calls: no GPU here)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header, prefix):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(%s_[a-z0-9_]+)\s*\(" % prefix, text)))


def test_hip_library_exports_header(ha):
    names = _declared("hanamaru_hip.h", "hr")
    assert len(names) >= 29 and "hr_comm_info" in names and "hr_accumulator_sum" in names
    debug = _declared("hanamaru_hip_debug.h", "hr")
    assert set(debug) == {"hr_set_debug_option", "hr_debug_draws", "hr_debug_path_draws", "hr_debug_path_draw_residuals", "hr_debug_path_log", "hr_debug_intersect", "hr_debug_trace", "hr_debug_wf_profile"}
    assert not [n for n in names if n.startswith("hr_debug_") or n == "hr_set_debug_option"], "the product header declares a debug entry point"
    lib = C.CDLL(ha.HIP_LIB)
    for n in names + debug:
        assert hasattr(lib, n), "libhanamaru_hip.so lacks %s" % n
    assert ha.hip_lib().hr_abi_version() == 7


def test_product_hosts_bind_the_product_header_only(ha):
    """hr_debug_* / hr_set_debug_option are declared in hanamaru_hip_debug.h, not in the product header: the hanamaru-hip CLI (the C++ stand-in
    for the Rust shell) must not import one of them, and neither the Rust mirror nor INTEGRATION.md may name one as something to bind."""
    import subprocess
    cli = os.path.join(ROOT, "hanamaru-renderer_amd", "hanamaru-hip")
    if not os.path.exists(cli):
        pytest.skip("CLI not built (needs libhanamaru_hip.so: __graft_entry__.build())")
    syms = subprocess.run(["nm", "-D", "--undefined-only", cli], stdout=subprocess.PIPE, text=True, check=True).stdout
    used = sorted(set(re.findall(r"\b(hr_[a-z0-9_]+)", syms)))
    assert used and "hr_render" in used
    assert not [n for n in used if n.startswith("hr_debug_") or n == "hr_set_debug_option"], used
    assert "hanamaru_hip_debug.h" not in open(os.path.join(ROOT, "hanamaru-renderer_amd", "host", "cli_main.cpp")).read()
    ffi = open(os.path.join(ROOT, "rust", "hip_ffi.rs")).read()
    assert not re.search(r"pub fn (hr_debug_|hr_set_debug_option)", ffi)


def test_host_library_exports_header(ha):
    names = _declared("hanamaru_host.h", "hh")
    assert len(names) >= 9
    lib = C.CDLL(ha.HOST_LIB)
    for n in names:
        assert hasattr(lib, n), "libhanamaru_host.so lacks %s" % n


def test_struct_layouts_match_header(ha):
    # sizes the C compiler gives the POD structs (include/hanamaru_hip.h) vs the ctypes mirrors
    assert C.sizeof(ha.Vec3) == 24 and C.sizeof(ha.Texture) == 32 and C.sizeof(ha.Material) == 16 + 3 * 32
    assert C.sizeof(ha.Image) == 16 and C.sizeof(ha.Camera) == 6 * 24 + 24 and C.sizeof(ha.Skybox) == 48
    assert C.sizeof(ha.Element) == 8 + 112 + 32 + 48 + 32
    assert C.sizeof(ha.Stats) == 46 * 8 and C.sizeof(ha.CommInfo) == 32


def test_no_device_is_a_clean_error(ha):
    """Without a GPU hr_create must fail with an error code and text — never crash, never fall back."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(ha.HipError) as e:
        ha.Renderer(0)
    assert e.value.code < 0 and str(e.value)


def test_product_does_not_reference_oracle_or_emulation():
    """The product tree must not import, link or name the checker (oracle/) or the host emulation (tests/emu)."""
    pkg = os.path.join(ROOT, "hanamaru-renderer_amd")
    offenders = []
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".cpp", ".h", ".hip", ".py", "Makefile")):
                text = open(os.path.join(d, f), errors="ignore").read()
                for line in text.splitlines():
                    code = line.split("//")[0]
                    if re.search(r"liboracle|oracle_py|libhr_emu|emu_py|#include\s+\"\.\./\.\./oracle", code):
                        offenders.append((f, line.strip()))
    assert not offenders, offenders


def test_graft_entry_build_check_follows_the_header():
    """__graft_entry__.build() compares the library's ABI version with the header's, not with a literal (a literal went stale once)."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "__graft_entry__.py")).read()
    assert "HR_ABI_VERSION" in src and not re.search(r"hr_abi_version\(\)\s*==\s*\d", src)



def test_rust_mirror_carries_the_headers_layout():
    """rust/hip_ffi.rs has never been compiled (no Rust toolchain here): its generated block of compile-time size / alignment assertions and
    its field-offset test must be the ones tools/gen_rust_layout.py derives from include/hanamaru_hip.h with the C compiler, so that the first
    `cargo build` fails loudly on a layout disagreement instead of mis-reading a scene."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    assert subprocess.run([sys.executable, os.path.join(root, "tools", "gen_rust_layout.py"), "--check"]).returncode == 0, "run tools/gen_rust_layout.py"
    src = open(os.path.join(root, "rust", "hip_ffi.rs")).read()
    assert "size_of::<HrElement>() == 232" in src and "pub const HR_ABI_VERSION: i32 = %d;" % int(re.search(r"#define\s+HR_ABI_VERSION\s+(\d+)", open(os.path.join(root, "include", "hanamaru_hip.h")).read()).group(1)) in src
