#!/bin/bash
# Build container: the variant libraries profiles/session_r06.sh compares with the product on one box (interleaved runs):
#   w4old  step_wide<4> as round 5 left it (no fused bin, round 5's output tail)      w4fma  ... with the fused bin only
#   w2old  step_wide<2> as round 5 left it (w2noguard: this round's, the far-entry guard off)                                            f64old step_fast64 with round 5's float32 screening
set -e
cd "$(dirname "$0")/.."
bash profiles/build_variant.sh w4old "-DDIRAL_WIDE_FIN_FMA=0 -DDIRAL_WIDE_P4V2=0" k_wide4
bash profiles/build_variant.sh w4fma "-DDIRAL_WIDE_P4V2=0" k_wide4
bash profiles/build_variant.sh w2old "-DDIRAL_WIDE_FIN_FMA=0 -DDIRAL_WIDE_P4V2=0" k_wide2
bash profiles/build_variant.sh w2noguard "-DDIRAL_WIDE_FAR_GUARD=0" k_wide2        # step_wide<2> of this round without the far-entry guard
bash profiles/build_variant.sh f64old "-DDIRAL_FAST_F32_FMA=0" k_fast64
