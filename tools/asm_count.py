"""Static instruction counts per kernel of a gfx950 assembly listing (hipcc -S --cuda-device-only): VALU / SALU / memory
instructions between a kernel's label and its .Lfunc_end.  A quick look at what a source change did to a kernel
before spending GPU time on it (static, not dynamic: loops and branches are not weighted)."""
import re
import sys


def main(path, *filters):
    text = open(path).read()
    for m in re.finditer(r"^(_Z\w+):.*?\n(.*?)^\.Lfunc_end\d+:", text, re.S | re.M):
        name, body = m.group(1), m.group(2)
        if filters and not any(f in name for f in filters):
            continue
        ins = [l.strip() for l in body.split("\n") if l.startswith("\t") and not l.strip().startswith((".", ";"))]
        v = sum(l.startswith("v_") for l in ins)
        s = sum(l.startswith("s_") for l in ins)
        g = sum(l.startswith(("global_", "buffer_", "flat_", "scratch_")) for l in ins)
        d = sum(l.startswith("ds_") for l in ins)
        print(f"{name[:70]:70s} valu {v:6d} salu {s:6d} vmem {g:5d} lds {d:5d}")


if __name__ == "__main__":
    main(*sys.argv[1:])
