#!/bin/bash
# Round-3 experiment batch A (timing only): forward LDS-exchange width, MFMA beside the VALU, lambda = O psi XCD map.
cd /root/repo
S=${1:-1024}
for opt in "" "observable_xcd_states=0"; do QHBM_OPTS=$opt python scripts/experiments/vqt_time.py $S; done
for v in base fwd_b128_store fwd_b128_load fwd_b128_both fwd_mfma16 fwd_mfma fwd_no_x fwd_no_x_mfma; do
  QHBM_ENGINE_LIB=scripts/experiments/ablate/lib_$v.so python scripts/experiments/vqt_time.py $S
done
