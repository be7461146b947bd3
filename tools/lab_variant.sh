#!/bin/bash
# build throw-away variants of the library into tools/lab/<name>/ (timing only):  tools/lab_variant.sh <name> <extra hipcc flags...>
# The source list is the Makefile's (SRCS), so a new .hip file cannot be forgotten here.
set -e
name=$1; shift
mkdir -p tools/lab/$name
srcs=$(sed -n 's/^SRCS  *:= *//p' Makefile | sed 's#\$(CSRC)#vita_amd/csrc#g')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -w "$@" -shared -o tools/lab/$name/libvita_hip.so $srcs
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -std=c++17 "$@" tools/bench_gemm.cpp -o tools/lab/$name/bench_gemm -Ltools/lab/$name -lvita_hip -Wl,-rpath,'$ORIGIN'
