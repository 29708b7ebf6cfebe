#!/bin/bash
# Registers / LDS / spills of the kernels of one object file of the library: tools/regs.sh <file.o> [name pattern]
o=$(realpath "$1"); pat=${2:-}; here=$(dirname $(realpath $0))
d=$(mktemp -d); cp $o $d/x.o; cd $d
/opt/rocm/lib/llvm/bin/llvm-objdump --offloading x.o > /dev/null
/opt/rocm/lib/llvm/bin/llvm-readelf --notes x.o.0.hipv4-amdgcn-amd-amdhsa--gfx950 > notes.txt
python3 $here/kernel_regs.py notes.txt "$pat"
rm -rf $d
