#!/bin/bash
# builds libflate_amd/liblfx_<tag>.so = the product objects with lfx_inflate_fast.hip recompiled under extra -D macros
tag=$1; shift
cd "$(dirname "$0")/../../libflate_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c lfx_inflate_fast.hip -o lfx_inflate_fast_$tag.o "$@" 2>&1 | grep -E "error" -A3
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../liblfx_$tag.so lfx_encode_kernels.o lfx_match7.o lfx_match5.o lfx_parse2.o lfx_decode_kernels.o lfx_inflate_fast_$tag.o lfx_api.o lfx_decode.o lfx_sharded.o lfx_hostio.o && ls -la ../liblfx_$tag.so
