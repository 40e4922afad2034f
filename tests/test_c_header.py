"""include/eds_hip.h is the drop-in boundary: plain C.  It must compile as C99 and as C++11 with warnings as errors, a C program that
takes the address of EVERY function it declares must link against libeds_hip.so, and the struct sizes the C compiler sees must be the
ones the library was built with (no GPU needed: nothing here launches anything)."""
import importlib
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "include", "eds_hip.h")
capi = importlib.import_module("slam-eds_amd.capi")


def _declared_functions():
    text = open(HDR).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return sorted(set(re.findall(r"\b(eds_[a-z0-9_]+)\s*\(", text)) - {"eds_trk", "eds_pyr"})


def test_header_is_c99_and_cxx11_clean(tmp_path):
    for lang, std, cc in (("c", "-std=c99", "gcc"), ("c++", "-std=c++11", "g++")):
        src = tmp_path / ("inc." + ("c" if lang == "c" else "cpp"))
        # both headers of the boundary: the tracker's C ABI and the RCCL gather of the result table (opaque pointers: no rccl.h needed to include it)
        src.write_text('#include "eds_hip.h"\n#include "eds_hip_rccl.h"\nint main(void) { return 0; }\n')
        subprocess.check_call([cc, std, "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", str(src),
                               "-o", str(tmp_path / "inc.o")])


def test_c_program_links_every_declared_function(tmp_path):
    capi.build()
    names = _declared_functions()
    assert len(names) >= 50 and set(capi.EXPORTS) <= set(names) | {"eds_last_error"}, (len(names), sorted(set(capi.EXPORTS) - set(names)))
    lines = ['#include <stdio.h>', '#include "eds_hip.h"', "int main(void) {", "    const void* f[] = {"]
    lines += [f"        (const void*)(size_t)&{n}," for n in names]
    lines += ["    };", "    size_t i, n = sizeof(f) / sizeof(f[0]);", "    for (i = 0; i < n; ++i) if (!f[i]) return 2;",
              "    if (eds_abi_version() != EDS_HIP_ABI_VERSION) return 3;",
              "    if (eds_trk_cfg_size() != (int)sizeof(eds_trk_cfg) || eds_trk_info_size() != (int)sizeof(eds_trk_info)) return 4;",
              '    printf("%d functions, cfg %d B, info %d B\\n", (int)n, (int)sizeof(eds_trk_cfg), (int)sizeof(eds_trk_info));',
              "    return 0;", "}"]
    src = tmp_path / "link.c"
    src.write_text("\n".join(lines) + "\n")
    libdir = os.path.dirname(capi.LIB_PATH)
    exe = tmp_path / "link"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                           "-L", libdir, "-leds_hip", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.check_output([str(exe)], text=True)
    assert "functions" in out
