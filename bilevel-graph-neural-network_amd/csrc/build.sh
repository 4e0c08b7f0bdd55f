#!/bin/bash
# Builds libblgnn.so for gfx950 (cross-compiles without a GPU).  Output stays in-tree
# (bilevel-graph-neural-network_amd/lib/) so that it travels to the GPU box.
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
out="$here/../lib"
mkdir -p "$out" "$here/obj"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function"
pids=()
for f in runtime graph_layout gather readout linear linear_x6 linear_h3 gat_transform gmn_match gmn_match_sp gin_fused gin_x6 gin_h3 linear_sp gin_stack gin_stack_h3 link_head backward gat_backward gmn_backward gru; do
  src="$here/$f.hip"; obj="$here/obj/$f.o"
  stale=0
  [[ -f "$obj" && "$src" -ot "$obj" ]] || stale=1
  for h in "$here"/*.hpp "$here/../../include/blgnn.h"; do [[ "$h" -nt "$obj" ]] && stale=1; done
  if [[ $stale -eq 1 ]]; then
    $HIPCC $FLAGS -c "$src" -o "$obj" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [[ -n "$p" ]] && wait "$p"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC "$here"/obj/*.o -o "$out/libblgnn.so"
echo "built $out/libblgnn.so"
