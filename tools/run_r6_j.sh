bash tools/run_r6_i.sh
bash tools/run_r6_h.sh
