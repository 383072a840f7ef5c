#!/usr/bin/env python3
"""Hash of the device sources (fidget_amd/csrc): ties profiles/traffic_*.json to the build it was measured on."""
import glob, hashlib, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_hash():
    h = hashlib.sha1()
    for f in sorted(glob.glob(os.path.join(ROOT, "fidget_amd", "csrc", "*"))):
        if os.path.isfile(f) and f.rsplit(".", 1)[-1] in ("py", "hip", "hpp", "h", "cpp"):
            h.update(os.path.basename(f).encode())
            h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(source_hash())
