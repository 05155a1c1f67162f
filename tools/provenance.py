"""What binary produced a number: SHA-256 of the library sources and of the built liblambda_snark_core.so, plus the git HEAD recorded
by __graft_entry__.build() (the .git directory does not travel to the GPU box).  bench.py prints it beside its roofline figures;
tools/prof_round3.sh stamps it into every profiles/r03_*.json; bench.py only quotes a profile's PMC traffic when the stamp matches
the library it is running."""
import glob
import hashlib
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "lambda-snark-r_amd", "csrc")
LIB = os.path.join(ROOT, "lambda-snark-r_amd", "lib", "liblambda_snark_core.so")
BUILD_INFO = os.path.join(ROOT, "lambda-snark-r_amd", "lib", "build_info.json")


def source_sha256():
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.hpp")) + glob.glob(os.path.join(CSRC, "*.cpp")) +
                   glob.glob(os.path.join(CSRC, "Makefile")) + glob.glob(os.path.join(ROOT, "include", "lambda_snark", "*.h")))
    for f in files:
        h.update(os.path.relpath(f, ROOT).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def file_sha256(path):
    if not os.path.exists(path):
        return None
    h = hashlib.sha256()
    with open(path, "rb") as fh:
        for block in iter(lambda: fh.read(1 << 20), b""):
            h.update(block)
    return h.hexdigest()


def provenance():
    info = {}
    if os.path.exists(BUILD_INFO):
        try:
            with open(BUILD_INFO) as f:
                info = json.load(f)
        except (OSError, ValueError):
            info = {}
    src = source_sha256()
    return {"source_sha256": src, "lib_sha256": file_sha256(LIB), "git_head": info.get("git_head"), "git_dirty": info.get("git_dirty"),
            "lib_built_from_these_sources": info.get("source_sha256") == src if info else None}


def matches(profile_json):
    """True when a profiles/*.json artefact was measured on the sources this tree holds."""
    stamp = (profile_json or {}).get("provenance") or {}
    return bool(stamp.get("source_sha256")) and stamp.get("source_sha256") == source_sha256()


if __name__ == "__main__":
    print(json.dumps(provenance()))
