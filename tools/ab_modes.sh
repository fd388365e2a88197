for rep in 1 2; do for m in "0 0" "1 1" "0 1" "1 0"; do set -- $m; MG_PW_MODE=$1 MG_RRW_MODE=$2 python bench.py --no-cpu-baseline --steps 30 2>/dev/null | python -c "
import json,sys
o=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('pw_mode', sys.argv[1], 'rrw_mode', sys.argv[2], round(o['ms_per_step'],4), 'pair', round(o['roofline']['launch_ms'],4), [round(k['launch_ms'],4) for k in o['kernels']])" $1 $2; done; done
