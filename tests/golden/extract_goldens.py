#!/usr/bin/env python3
"""Regenerates tests/golden/pixel_render_*.txt from the reference's own
integration test (fidget/tests/pixel_render.rs).  Run in the build container
only (the GPU box has no /root/reference); the outputs are committed.

Each `const EXPECTED*: &str = "..."` ASCII bitmap ('#' inside, '.' outside) is
written verbatim (leading indentation stripped) to
pixel_render_<fn>_<const>.txt, with the source line range in a header comment.
"""
import os
import re
import sys

SRC = "/root/reference/fidget/tests/pixel_render.rs"
OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    text = open(SRC).read()
    lines = text.split("\n")
    fn = None
    i = 0
    n = 0
    while i < len(lines):
        m = re.match(r"fn (check_\w+)<", lines[i])
        if m:
            fn = m.group(1)
        m = re.match(r'\s*const (EXPECTED\w*): &str = "$', lines[i])
        if m:
            name = m.group(1)
            start = i + 1
            rows = []
            i += 1
            while True:
                row = lines[i].strip()
                if row.endswith('";'):
                    rows.append(row[:-2])
                    break
                rows.append(row)
                i += 1
            path = os.path.join(OUT, f"pixel_render_{fn}_{name}.txt")
            with open(path, "w") as f:
                f.write(f"# fidget/tests/pixel_render.rs:{start + 1}-{i + 1} ({fn}::{name})\n")
                f.write("\n".join(rows) + "\n")
            n += 1
        i += 1
    print(f"wrote {n} golden bitmaps", file=sys.stderr)


if __name__ == "__main__":
    main()
