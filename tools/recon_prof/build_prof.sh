#!/bin/bash
# profiling variant of the library: intra.hip with -DVVC355_RECON_PROF, everything else from the regular build
set -e
cd /root/repo/ffvvc_amd/csrc
make -s
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fvisibility=default -ffp-contract=off -DVVC355_RECON_PROF -c intra.hip -o ../../tools/recon_prof/intra_prof.o
objs=$(ls *.o | grep -v '^intra.o$')
/opt/rocm/bin/hipcc -shared --offload-arch=gfx950 -o ../../tools/recon_prof/libvvc_mi355_prof.so ../../tools/recon_prof/intra_prof.o $objs
