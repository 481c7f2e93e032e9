#!/bin/bash
# builds quad_loop_<name> for a set of schedules / ablations (run in scripts/ubench; binaries are git-ignored)
set -e
NFD=""
cd "$(dirname "$0")"
build() { name=$1; shift; python gen_quad_loop.py "$@" > quad_loop.inc; hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -Wno-inline-asm $NFD quad_loop.hip -o quad_loop_$name; }
build s0 --sched 3,3,3,3,2,2,0,0
build s0st --sched 3,3,3,3,2,2,0,0 --stores 4
NFD="-DQUAD_NF=4"
build d0 --nf 4 --sched 4,4,4,0
build d1 --nf 4 --sched 3,3,3,3
build d2 --nf 4 --sched 4,4,2,2
build d0st --nf 4 --sched 4,4,4,0 --stores 2
build dmfma --nf 4 --sched 4,4,4,0 --no-reads --no-dma --no-barrier
build dnodma --nf 4 --sched 4,4,4,0 --no-dma
build3() { name=$1; shift; python gen_duo3_loop.py "$@" > quad_loop.inc; hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -Wno-inline-asm -DQUAD_NF=4 quad_loop.hip -o quad_loop_$name; }
build3 t0 --sched 4,4,4,0
build3 t1 --sched 3,3,3,3
build3 t2 --sched 2,2,4,4
build3 t0st --sched 4,4,4,0 --stores 2
build3 tnodma --sched 4,4,4,0 --no-dma
python gen_quad_loop.py > quad_loop.inc
