#!/usr/bin/env python3
"""Register / scratch / LDS use of every kernel in libmf_hip.so, read from the code objects' notes.

hipcc embeds one clang offload bundle per translation unit in the ``.hip_fatbin`` section; each bundle holds the gfx950
code object whose ``NT_AMDGPU_METADATA`` note lists, per kernel, ``.vgpr_count``, ``.agpr_count``, ``.sgpr_count``,
``.vgpr_spill_count``, ``.sgpr_spill_count``, ``.private_segment_fixed_size`` (scratch bytes per lane) and
``.group_segment_fixed_size`` (static LDS).  No GPU needed: ``tests/test_host_cpu.py`` asserts from this that no kernel
of the product paths spills.

    python tools/kernel_resources.py [path/to/libmf_hip.so] [--spills]
"""
from __future__ import annotations

import pathlib
import re
import struct
import subprocess
import sys
import tempfile

LLVM = pathlib.Path("/opt/rocm/lib/llvm/bin")
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
ROOT = pathlib.Path(__file__).resolve().parents[1]
DEFAULT_LIB = ROOT / "matrix-factorization-torch_amd" / "lib" / "libmf_hip.so"
FIELDS = ("vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size",
          "group_segment_fixed_size", "max_flat_workgroup_size")


def code_objects(lib: pathlib.Path) -> list[bytes]:
    """The gfx950 code objects of every bundle in the library's .hip_fatbin section."""
    with tempfile.TemporaryDirectory() as tmp:
        fat = pathlib.Path(tmp) / "fat.bin"
        subprocess.run([str(LLVM / "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", str(lib), str(fat)], check=True)
        data = fat.read_bytes()
    out = []
    for m in re.finditer(re.escape(MAGIC), data):
        base = m.start()
        (n,) = struct.unpack_from("<Q", data, base + len(MAGIC))
        pos = base + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", data, pos)
            triple = data[pos + 24: pos + 24 + tlen].decode()
            pos += 24 + tlen
            if "gfx950" in triple and size:
                out.append(data[base + off: base + off + size])
    return out


def demangle(names: list[str]) -> list[str]:
    try:
        res = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True)
        out = res.stdout.splitlines()
        return out if len(out) == len(names) else names
    except (OSError, subprocess.CalledProcessError):
        return names


def kernel_resources(lib: pathlib.Path = DEFAULT_LIB) -> dict[str, dict[str, int]]:
    """{demangled kernel name: {field: value}} for every kernel of the library."""
    table: dict[str, dict[str, int]] = {}
    for co in code_objects(pathlib.Path(lib)):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co)
            f.flush()
            notes = subprocess.run([str(LLVM / "llvm-readelf"), "--notes", f.name], capture_output=True, text=True, check=True).stdout
        # one YAML map per kernel under amdhsa.kernels: entries start with "  - .agpr_count:" (keys sorted) -- split on list items
        body = notes.split("amdhsa.kernels:", 1)[1] if "amdhsa.kernels:" in notes else ""
        body = body.split("amdhsa.target:", 1)[0]
        for item in re.split(r"\n  - ", "\n" + body)[1:]:
            name = re.search(r"\.name:\s+(\S+)", item)
            if not name:
                continue
            rec = {}
            for fld in FIELDS:
                mm = re.search(rf"\.{fld}:\s+(\d+)", item)
                rec[fld] = int(mm.group(1)) if mm else 0
            table[name.group(1)] = rec
    names = list(table)
    return dict(zip(demangle(names), table.values()))


def main() -> None:
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    res = kernel_resources(pathlib.Path(args[0]) if args else DEFAULT_LIB)
    only_spills = "--spills" in sys.argv
    print(f"{'vgpr':>5} {'agpr':>5} {'sgpr':>5} {'vspill':>6} {'sspill':>6} {'scratch':>7} {'lds':>7}  kernel")
    for name, r in sorted(res.items()):
        if only_spills and not (r["vgpr_spill_count"] or r["private_segment_fixed_size"]):
            continue
        print(f"{r['vgpr_count']:5d} {r['agpr_count']:5d} {r['sgpr_count']:5d} {r['vgpr_spill_count']:6d} {r['sgpr_spill_count']:6d} "
              f"{r['private_segment_fixed_size']:7d} {r['group_segment_fixed_size']:7d}  {name[:150]}")


if __name__ == "__main__":
    main()
