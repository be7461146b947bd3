#!/bin/bash
# like lab_variant.sh, but recompiles only the named sources with the extra flags and links the main build's objects for the rest:
#   tools/lab_variant_fast.sh <name> "<file.hip> [<file.hip> ...]" <extra hipcc flags...>
set -e
name=$1; files=$2; shift 2
mkdir -p tools/lab/$name
objs=""
for src in $(sed -n 's/^SRCS  *:= *//p' Makefile | sed 's#\$(CSRC)#vita_amd/csrc#g'); do
  base=$(basename $src .hip)
  if echo " $files " | grep -q " $base.hip "; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -w "$@" -c $src -o tools/lab/$name/$base.o &
    objs="$objs tools/lab/$name/$base.o"
  else
    objs="$objs vita_amd/csrc/$base.o"
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/lab/$name/libvita_hip.so $objs
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -std=c++17 "$@" tools/bench_gemm.cpp -o tools/lab/$name/bench_gemm -Ltools/lab/$name -lvita_hip -Wl,-rpath,'$ORIGIN'
