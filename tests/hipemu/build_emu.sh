#!/bin/bash
# TEST INFRASTRUCTURE ONLY: compiles the UNMODIFIED product sources (typesense_amd/csrc/*.hip) for the host CPU
# against tests/hipemu/hip/hip_runtime.h (SIMT emulator) -> tests/hipemu/_build/libtsgpu_emu.so.
# Used by the `not gpu` tests to exercise kernel + planning logic where no GPU exists. Never shipped, never
# loaded by typesense_amd (which only ever loads the real libtsgpu.so and fails loudly without it).
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
CXX=/opt/rocm/lib/llvm/bin/clang++
[ -x "$CXX" ] || CXX=clang++
mkdir -p "$HERE/_build"
# optional: $1 = extra -D flags, $2 = output suffix (variant builds for tests that force rarely-taken paths)
EXTRA="$1"
OUT="$HERE/_build/libtsgpu_emu$2.so"
SRCS="$ROOT/typesense_amd/csrc/tsgpu.hip $ROOT/typesense_amd/csrc/tsgpu_index.hip $ROOT/typesense_amd/csrc/tsgpu_vec.hip $ROOT/typesense_amd/csrc/tsgpu_facet.hip $ROOT/typesense_amd/csrc/tsgpu_group.hip"
# one builder at a time (pytest -n: every worker calls this), and the library appears atomically
exec 9>"$OUT.lock"
flock 9
NEWER=0
for f in $SRCS "$ROOT"/typesense_amd/csrc/*.h "$ROOT"/include/*.h "$HERE"/hip/hip_runtime.h; do
  if [ ! -f "$OUT" ] || [ "$f" -nt "$OUT" ]; then NEWER=1; fi
done
if [ "$NEWER" = 1 ]; then
  "$CXX" -x c++ -std=c++17 -O1 -g -fPIC -shared -ffp-contract=off -DTSGPU_HIP_EMU=1 -Wno-unused-value -Wno-macro-redefined -Wno-psabi \
      $EXTRA -I "$HERE" -o "$OUT.tmp.$$" $SRCS -lpthread -ldl
  mv -f "$OUT.tmp.$$" "$OUT"
fi
echo "$OUT"
