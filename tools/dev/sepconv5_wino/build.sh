#!/bin/bash
# PROTOTYPE build: tools/dev/sepconv5_wino/libwino15.so (not part of libpcfa_hip.so)
cd "$(dirname "$0")" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC proto.hip -o libwino15.so
