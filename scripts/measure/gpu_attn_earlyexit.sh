for L in 256 384 512; do echo "== ENC_LEN=90,250 ENC_L=$L"; ENC_LEN=90,250 ENC_L=$L bash scripts/measure/gpu_attn_ab.sh libmdrhip.so; done
