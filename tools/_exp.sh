cd $GRAFT_REPO_ROOT
for sh in qkv fc1; do python tools/gemm_timeline.py --shape $sh --tile 15,25 --waits; done
for sh in w13 lin1; do python tools/gemm_timeline.py --shape $sh --tile 15,25 --waits --conv; done
python tools/gemm_timeline.py --shape w2 --tile 15 --ksplit 5 --partials --waits --conv
python tools/gemm_timeline.py --shape w13 --tile 19,29 --waits --conv --m 4000
python tools/gemm_timeline.py --shape w13 --tile 15 --brief --conv
python tools/gemm_timeline.py --shape qkv --tile 15 --brief
