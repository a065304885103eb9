# A/B bench on the GPU box: tools/ab.sh TAG "VARIANT_LIB_NAMES..."   (variants built by tools/build_variant.sh; "main" = the library in tree)
# runs the GPU parity suite first, then configs[2] twice per variant, configs[1] and one rank of an 8-way split once.
cd $GRAFT_REPO_ROOT
TAG=$1; shift
timeout -k 10 500 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_tests.log 2>&1 || { tail -30 gpurun_out/${TAG}_tests.log; exit 1; }
tail -2 gpurun_out/${TAG}_tests.log
V=$GRAFT_REPO_ROOT/pathtrace-on-cuda_amd/build
run() { n=$1; shift; if [ $n = main ]; then L=X=1; else L=PTAMD_LIB=$V/libptamd_$n.so; fi; echo "== $n $*"; env $L timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-probes "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']; print(round(d['value'],1), 'ms/step', round(d['ms_per_step'],1), 'trace sum', round(r['kernel_ms_sum'],1), 'iters', r['bounce_iterations'])"; }
for rep in 1 2; do for n in "$@"; do run $n; done; done
for n in "$@"; do run $n --config 1; done
for n in "$@"; do run $n --emulate-world 8 --rank 0; done
