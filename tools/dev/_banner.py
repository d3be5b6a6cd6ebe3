"""Imported by every dev script that launches kernels: the FIRST line such a script prints is the identity of the binary it runs -
nmpc_version(): ABI, hash of the kernel sources, which code generation runs by default.  (Round 3 lost the one GPU fault this repository has
recorded because the script in flight did not say which build it was running: the evidence that would have named the instruction was thrown
away by process, not by the machine.)"""
from rotors_mpc_controller_amd import _lib

print("# " + _lib.load().nmpc_version().decode(), flush=True)
