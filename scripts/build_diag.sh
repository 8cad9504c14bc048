#!/bin/bash
# the measurement build (-DTBX_DIAG: parts of kernels / launch paths switchable by environment variables; never the shipped library)
# into scripts/ab/lib_diag.so, objects in /tmp so that the product build's objects stay untouched
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
O=/tmp/tbx_diag_objs; mkdir -p $O $R/scripts/ab
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -DTBX_DIAG -Wno-unused-value"
pids=""
for f in engine agent breakout space_invaders amidar gridworld gather; do
  if [ ! -f $O/$f.o ] || [ -n "$(find $R/toybox_amd/csrc $R/include -newer $O/$f.o \( -name '*.hip' -o -name '*.hpp' -o -name '*.h' \) | head -1)" ]; then
    /opt/rocm/bin/hipcc $FLAGS -c -o $O/$f.o $R/toybox_amd/csrc/$f.hip & pids="$pids $!"
  fi
done
for p in $pids; do wait $p; done
/opt/rocm/bin/hipcc $FLAGS -shared -o $R/scripts/ab/lib_diag.so $O/engine.o $O/agent.o $O/breakout.o $O/space_invaders.o $O/amidar.o $O/gridworld.o $O/gather.o -ldl
ls -la $R/scripts/ab/lib_diag.so
