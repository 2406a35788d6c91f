#!/bin/bash
# Regenerates tests/golden/philox_kat_rocrand.json from rocRAND's host-callable engine.
set -euo pipefail
here="$(cd "$(dirname "$0")" && pwd)"
out="$here/../_build"; mkdir -p "$out"
hipcc -O1 -o "$out/gen_philox_kat" "$here/gen_philox_kat.cpp"
"$out/gen_philox_kat" > "$here/../../tests/golden/philox_kat_rocrand.json"
echo "wrote tests/golden/philox_kat_rocrand.json"
