# lanes-per-chain heuristic: the resident-waves-per-CU target (SAT_EXP_LPC_WAVES, default 8) on the large-entry workloads
for c in c4 q101 n96; do for w in 8 12 16; do echo -n "$c target=$w: "; SAT_EXP_LPC_WAVES=$w timeout -k 10 200 python scripts/run_config.py $c 3 2>&1 | tail -1 | sed 's/.*search -> //; s/scorings.*//'; done; done
