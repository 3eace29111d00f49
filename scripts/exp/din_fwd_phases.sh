for s in 1 2 3 4 0; do echo "stop=$s"; REC_DIN_STOP=$s timeout -k 10 120 python3 scripts/exp/din_attn_time.py 2>&1 | grep "V= 5"; done
