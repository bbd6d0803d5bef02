cd $GRAFT_REPO_ROOT
run() { python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-live-traffic 2>/dev/null | tail -1 | python -c "
import sys,json; j=json.loads(sys.stdin.read()); r=j['roofline']; print('$1 step %.2f ms walk %.2f ms' % (j['ms_per_step'], r['avg_launch_ms']))"; }
run default
MPG_SPLIT_SLICE=4194304 run slice4M_ov0
MPG_SPLIT_OVERLAP=1 MPG_SPLIT_SLICE=4194304 run slice4M_ov1
MPG_SPLIT_OVERLAP=1 MPG_SPLIT_SLICE=2097152 run slice2M_ov1
MPG_SPLIT_OVERLAP=1 MPG_SPLIT_SLICE=1048576 run slice1M_ov1
MPG_SPLIT_OVERLAP=1 MPG_SPLIT_SLICE=524288 run slice512k_ov1
run default
