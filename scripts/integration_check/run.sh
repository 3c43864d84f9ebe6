#!/bin/bash
# Compile-only check of the drop-in shim (INTEGRATION.md sections 2-3) against athena's own module sources, READ IN PLACE
# from the reference checkout (never copied), plus compile-only stand-ins for the three libraries this image lacks.
# Outputs go to build/integration_check/ (git-ignored).  Nothing is linked or run: syntax / interface evidence only.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
R="${ATHENA_REFERENCE:-/root/reference}/src/athena"
FC="${FC:-$(command -v amdflang || echo /opt/rocm/bin/amdflang)}"
[ -d "$R" ] || { echo "reference checkout not found at $R"; exit 3; }
[ -x "$FC" ] || { echo "no Fortran compiler"; exit 3; }
OUT="$ROOT/build/integration_check"
rm -rf "$OUT"; mkdir -p "$OUT"; cd "$OUT"
"$FC" -cpp -c "$HERE/stubs.f90" -o stubs.o
for f in athena_misc_types athena_diffstruc_extd athena_clipper athena_base_layer athena_msgpass_layer; do
  "$FC" -cpp -c "$R/$f.f90" -o "$f.o"          # athena's real modules (interfaces; bodies live in submodules)
done
"$FC" -cpp -c "$ROOT/athena_amd/fortran/athena_mp_c.f90" -o athena_mp_c.o
"$FC" -cpp -c "$HERE/hip_duvenaud_gno_ops.f90" -o hip_duvenaud_gno_ops.o   # the autodiff ops of all three layers
ls athena_mp__hip_ops.mod >/dev/null
"$FC" -cpp -c "$HERE/hip_kipf_msgpass.f90" -o hip_kipf_msgpass.o           # the Kipf layer TYPE
ls athena_mp__hip_kipf.mod >/dev/null
# the two layer TYPES that extend(msgpass_layer_type): Duvenaud (update_message + update_readout, the fused entry points bound
# inside the tape) and graph_nop (one-call reverse pass, forwarded edge geometry) -- INTEGRATION.md section 3
"$FC" -cpp -c "$HERE/hip_duvenaud_gno_layers.f90" -o hip_duvenaud_gno_layers.o
ls athena_mp__hip_layers.mod >/dev/null
echo "integration shim compiles against athena__msgpass_layer / athena__base_layer / diffstruc surface: OK"
