#!/bin/bash
# Diagnostic build of the library with in-kernel cycle stamps in nn.hip and train.hip (k_conv_t; train_net.hip shares the handle's layout) (-DDBAZ_STAMP; EXTRA_DEFS=-DDBAZ_DEBUG adds the A/B variants), into build/stamp/ (not shipped,
# git-ignored; it travels to the GPU box with the snapshot).  Use with DBAZ_LIB=$PWD/build/stamp/libdbaz_hip.so.
set -e
cd "$(dirname "$0")/.."
mkdir -p build/stamp
python -m dotsboxesaz_amd.build > /dev/null
for f in engine nn train train_net; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -DDBAZ_STAMP $EXTRA_DEFS -c dotsboxesaz_amd/csrc/$f.hip -o build/stamp/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/stamp/libdbaz_hip.so dotsboxesaz_amd/csrc/tree.o build/stamp/engine.o build/stamp/nn.o dotsboxesaz_amd/csrc/replay.o build/stamp/train.o build/stamp/train_net.o dotsboxesaz_amd/csrc/buildinfo.o
echo build/stamp/libdbaz_hip.so
