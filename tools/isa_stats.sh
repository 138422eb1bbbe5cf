#!/bin/bash
# Static instruction mix of one megakernel instantiation (default: the headline FLAT kernel): tools/isa_stats.sh [mangled-name-substring] [extra hipcc flags]
pat=${1:-megakernelILi0ELb0ELb0ELb1ELb0ELb1ELb1}; shift
root=$(cd "$(dirname "$0")/.." && pwd)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize "$@" -S --cuda-device-only -o /tmp/pt_isa.s $root/cudapathtracer_amd/csrc/pt_mk_lds.hip 2>/dev/null
awk "/^_ZN2pt[0-9]*${pat}[A-Za-z0-9_]*:/,/s_endpgm/" /tmp/pt_isa.s > /tmp/pt_isa_fn.s
echo "VALU $(grep -c '^\s*v_' /tmp/pt_isa_fn.s)  SALU $(grep -c '^\s*s_' /tmp/pt_isa_fn.s)  ds $(grep -c 'ds_' /tmp/pt_isa_fn.s)  global_load $(grep -c global_load /tmp/pt_isa_fn.s)  flat $(grep -c 'flat_' /tmp/pt_isa_fn.s)  scratch $(grep -c scratch_ /tmp/pt_isa_fn.s)  lane-spill $(grep -c 'v_readlane\|v_writelane' /tmp/pt_isa_fn.s)  div $(( $(grep -c v_div_scale /tmp/pt_isa_fn.s) / 2 ))  sqrt $(grep -c 'v_sqrt' /tmp/pt_isa_fn.s)"
grep "$pat" -A60 /tmp/pt_isa.s | grep -E "^; (TotalNumSgprs|NumVgprs|ScratchSize|Occupancy|LDSByteSize)" | head -5 | tr '\n' ' '; echo
