#!/bin/bash
# A/B builds of ONE kernel file: scripts/mkvar_file.sh FILE NAME "FLAGS" -> pt-three-ways_amd/libptw_hip_pw$NAME.so
# (csrc/FILE.hip recompiled with FLAGS, every other object from the `make` of the tree; select with PTW_LIB_PATH)
set -e
cd "$(dirname "$0")/../pt-three-ways_amd"
hipcc --offload-arch=gfx950 -std=c++17 -O3 -fPIC -Icsrc $3 -c csrc/$1.hip -o csrc/$1.pw$2.o
OBJS=""
for o in host/scene_builder.o host/obj_loader.o host/scenes.o host/framebuffer.o host/precompute.o host/bvh.o host/prefilter.o csrc/capi_host.o csrc/dispatch.o csrc/seq_single.o csrc/seq_spec.o csrc/seq_worker.o csrc/seq_worker2.o csrc/seq_worker_pre.o csrc/seq_worker2_pre.o csrc/perpixel.o csrc/accel.o csrc/resolve_kat.o csrc/capi_render.o csrc/capi_comm.o; do
  if [ "$o" = "csrc/$1.o" ]; then OBJS="$OBJS csrc/$1.pw$2.o"; else OBJS="$OBJS $o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o libptw_hip_pw$2.so -ldl -lpthread
echo built libptw_hip_pw$2.so
