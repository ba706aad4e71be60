#!/bin/bash
# A diagnostic / ablation BUILD of the library beside the product: objects and the .so go to <dir> (under /tmp), the in-tree library is
# not touched, so a measurement that dies half-way leaves the product as it was.  Load it with RISP_HIP_LIBRARY=<dir>/libreconfigisp_hip.so.
# usage: tools/build_variant.sh <dir> "<-D flags>" <source.hip> [...]     (the named sources are the ones the flags reach; the other objects
#        are copied from the in-tree build when they are there; `all` = every source)
set -eu
DIR=$1; FLAGS=$2; shift 2
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$DIR"
if ls "$REPO"/build/*.o > /dev/null 2>&1; then cp -p "$REPO"/build/*.o "$DIR"/; fi
for f in "$@"; do if [ "$f" = all ]; then rm -f "$DIR"/*.o; else rm -f "$DIR/$f.o"; fi; done
make -s -C "$REPO/reconfigisp_amd/csrc" -j8 OUT="$DIR/libreconfigisp_hip.so" OBJDIR="$DIR" EXTRA="$FLAGS" > "$DIR/build.log" 2>&1 || { tail -20 "$DIR/build.log" >&2; exit 1; }
echo "$DIR/libreconfigisp_hip.so"
