"""Build bookkeeping of libaehmc_hip.so: the library is stamped with a hash of the sources it was
compiled from, so that "is the binary the tree's?" is decided by CONTENT, not by file times (a `.so`
copied onto another box, a header edited after the build, a checkout that resets mtimes)."""
from __future__ import annotations

import hashlib
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
CSRC = os.path.join(_HERE, "csrc")
INCLUDE = os.path.join(ROOT, "include")
LIB = os.path.join(_HERE, "libaehmc_hip.so")
STAMP = LIB + ".srchash"


def source_files():
    for d in (CSRC, INCLUDE):
        if not os.path.isdir(d):  # (an installed copy without the repository's include/ or csrc/ directory)
            raise RuntimeError(f"aehmc_amd: {d} is missing -- the library is built from, and checked against, the sources "
                               "in aehmc_amd/csrc and include/ of its repository checkout (run from the checkout, or set "
                               "AEHMC_AMD_LIB to a library you built yourself)")
    files = [os.path.join(CSRC, f) for f in os.listdir(CSRC)
             if f.endswith((".hip", ".cuh", ".h", ".inc")) or f == "Makefile"]
    files += [os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE) if f.endswith(".h")]
    return sorted(files)


def source_hash() -> str:
    """sha256 over (relative path, content) of every file the library is compiled from."""
    h = hashlib.sha256()
    for path in source_files():
        h.update(os.path.relpath(path, ROOT).encode() + b"\0")
        with open(path, "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    return h.hexdigest()


def stamped_hash():
    try:
        with open(STAMP) as f:
            return f.read().strip()
    except OSError:
        return None


def library_hash(path: str = LIB) -> str:
    """sha256 of the shared library itself (profiles/ summaries are tied to the binary they measured)."""
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for block in iter(lambda: f.read(1 << 20), b""):
            h.update(block)
    return h.hexdigest()


def is_current() -> bool:
    return os.path.exists(LIB) and stamped_hash() == source_hash()


def write_stamp() -> None:
    with open(STAMP, "w") as f:
        f.write(source_hash() + "\n")


def build(force: bool = False, quiet: bool = True) -> bool:
    """Compile the library unless the stamped source hash equals the tree's.  Returns True if it compiled.
    (`make` itself stamps the library -- csrc/Makefile runs `_build.py --stamp` after linking -- so a plain
    `make -C aehmc_amd/csrc -j8` gives a loadable library too.)"""
    want = source_hash()
    if not force and os.path.exists(LIB) and stamped_hash() == want:
        return False
    jobs = str(min(8, os.cpu_count() or 1))
    # -B: the decision to compile was taken on content; make's file times must not overrule it
    cmd = ["make", "-C", CSRC, "-B", "-j", jobs] + (["-s"] if quiet else [])
    subprocess.check_call(cmd)
    write_stamp()
    return True


if __name__ == "__main__":
    import sys
    if "--stamp" in sys.argv:
        write_stamp()
    else:
        print("compiled" if build(force="--force" in sys.argv, quiet=False) else "up to date")
