#!/bin/bash
# Regenerates profiles/r04_fault_emulation.txt: the launch that faulted in round 3 (gpurun_out/qp_check2.log: k_team_qp<per-stage, trajectories>,
# sim_num_steps = 4, qp_polish = 0, B = 256, aggressive seed 8) executed on the CPU by the ISA emulator, all 256 workgroups, for
#   (1) the sources of d4e0de2 - the last commit before the fault - built WITH -amdgpu-mfma-vgpr-form (how every kernel was built that hour),
#   (2) the same sources with the default code generation,
#   (3) the sources of this tree built with the flag, cold and warm-started,
# plus the static hazard-distance scan of the flag builds.  CPU only; ~15 minutes on 8 cores.
set -e
cd "$(dirname "$0")/../.."
OUT=profiles/r04_fault_emulation.txt
OLD=/tmp/nmpc_d4e0de2
rm -rf $OLD && mkdir -p $OLD && git archive d4e0de2 rotors_mpc_controller_amd/csrc include | tar -x -C $OLD
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -Wno-unused-function --cuda-device-only -S"
( cd $OLD/rotors_mpc_controller_amd/csrc && /opt/rocm/bin/hipcc $FL -mllvm -amdgpu-mfma-vgpr-form -o $OLD/flag.s nmpc_as.hip 2>/dev/null ) &
( cd $OLD/rotors_mpc_controller_amd/csrc && /opt/rocm/bin/hipcc $FL -o $OLD/noflag.s nmpc_as.hip 2>/dev/null ) &
make -s -C rotors_mpc_controller_amd/csrc build/nmpc_qp_flag.s build/nmpc_as.s >/dev/null 2>&1 &
wait
K=k_team_qpILb0ELb1EdEE
H="--headers $OLD/rotors_mpc_controller_amd/csrc --include $OLD/include"
{
  echo "# $(date -u +%F) hipcc: $(/opt/rocm/bin/hipcc --version | grep 'HIP version')"
  echo "# emulated launch: $K, sim_num_steps 4, qp_polish 0, per-stage linearisation, trajectories, B = 256 (one instance per wave), aggressive seed 8"
  echo "## (1) sources of d4e0de2, built with -mllvm -amdgpu-mfma-vgpr-form"
  python tools/emu/sweep_workgroups.py $OLD/flag.s $K 0 256 $H
  echo "## (2) sources of d4e0de2, default code generation"
  python tools/emu/sweep_workgroups.py $OLD/noflag.s $K 0 256 $H
  echo "## (3) this tree ($(python tools/source_hash.py)), built with the flag: cold, then warm-started from the oracle's first solve"
  python tools/emu/sweep_workgroups.py rotors_mpc_controller_amd/csrc/build/nmpc_qp_flag.s $K 0 256
  python tools/emu/sweep_workgroups.py rotors_mpc_controller_amd/csrc/build/nmpc_qp_flag.s $K 0 256 --warm
  echo "## static hazard distances (tools/emu/isa_checks.py)"
  python tools/emu/isa_checks.py $OLD/flag.s $K
  python tools/emu/isa_checks.py rotors_mpc_controller_amd/csrc/build/nmpc_qp_flag.s $K
  python tools/emu/isa_checks.py rotors_mpc_controller_amd/csrc/build/nmpc_as.s k_team_asILb1ELb0ELi1EdEE
  python tools/emu/isa_checks.py rotors_mpc_controller_amd/csrc/build/nmpc_as.s k_team_asILb0ELb1ELi1EdEE
} > $OUT 2>&1
tail -30 $OUT
