# REC_DIN_STOP (diagnostics): 12 = gather + barriers only, 13 = + steps (1)-(3), 14 = + step (4) gEff, 0 = everything
for s in 12 13 14 0; do echo "stop=$s"; REC_DIN_STOP=$s timeout -k 10 120 python3 scripts/exp/din_attn_time.py 2>&1 | grep "V= 5"; done
