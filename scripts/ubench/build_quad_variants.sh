#!/bin/bash
# builds quad_loop_<name> for a set of schedules / ablations (run in scripts/ubench; binaries are git-ignored)
set -e
cd "$(dirname "$0")"
build() { name=$1; shift; python gen_quad_loop.py "$@" > quad_loop.inc; hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -Wno-inline-asm quad_loop.hip -o quad_loop_$name; }
build s0 --sched 3,3,3,3,2,2,0,0
build s1 --sched 4,4,4,4,0,0,0,0
build s2 --sched 2,2,2,2,2,2,2,2
build s3 --sched 2,2,2,2,3,3,2,0
build s4 --sched 1,2,2,2,2,2,3,2
build nomfma --no-mfma
build noreads --no-reads
build nodma --no-dma
build nobar --no-barrier
build mfmaonly --no-reads --no-dma --no-barrier
python gen_quad_loop.py > quad_loop.inc
