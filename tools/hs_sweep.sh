# development aid: A/B of k_mid's launch shape on the headline scene (run on the GPU box)
for cfg in "0 1024 3" "0 1728 3" "1 1728 3" "1 1728 4" "1 1728 2" "1 1536 3" "1 2048 3"; do set -- $cfg; echo "HS=$1 N_SOLVE=$2 BUDGET=$3"; TJ_PAIR_HEAD_START=$1 TJ_N_SOLVE=$2 TJ_HS_BUDGET=$3 python bench.py --no-cpu --no-extra 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(j['ms_per_step'],5), {k[2:]:round(v*1e3,1) for k,v in j['roofline']['kernel_ms_per_launch'].items() if v and k!='k_begin'})"; done
