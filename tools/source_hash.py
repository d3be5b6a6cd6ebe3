#!/usr/bin/env python3
"""sha1 over the kernel sources (rotors_mpc_controller_amd/csrc/*.h*: .hpp and .hip, names and contents, sorted) and the Makefile that
carries their compiler flags: the Makefile compiles it into nmpc_version(), bench.py and tools/summarize_pmc.py compare it with the sources they see - a stale binary or a
stale profile is then visible instead of carrying a fresh hash."""
import hashlib
import sys
from pathlib import Path

CSRC = Path(__file__).resolve().parent.parent / "rotors_mpc_controller_amd" / "csrc"


def source_hash(csrc: Path = CSRC) -> str:
    h = hashlib.sha1()
    for f in sorted(csrc.glob("*.h*")) + [csrc / "Makefile"]:
        h.update(f.name.encode())
        h.update(f.read_bytes())
    return h.hexdigest()[:12]


if __name__ == "__main__":
    print(source_hash(Path(sys.argv[1]) if len(sys.argv) > 1 else CSRC))
