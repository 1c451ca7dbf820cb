#!/usr/bin/env python3
"""Hash of the sources libart_hip.so is built from (csrc/* + include/art_hip.h): the identity of a build.  Written into
every profiles/*.json by tools/summarize_profile.py and compared by bench.py before it quotes a profile's byte counts, so
that a kernel change without a re-profile cannot yield a stale roofline fraction (VERDICT r2 #5a).
    python tools/source_hash.py      prints the 16-hex-digit hash of the tree"""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = ["attosecondraytracing_amd/csrc/art_device.h", "attosecondraytracing_amd/csrc/art_kernels.hip",
         "attosecondraytracing_amd/csrc/art_scene.h", "include/art_hip.h"]


def source_hash(root=ROOT):
    h = hashlib.sha256()
    for f in FILES:
        h.update(f.encode() + b"\0")
        h.update(open(os.path.join(root, f), "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(source_hash())
