cd $GRAFT_REPO_ROOT
for sh in lin1 w2; do
python tools/gemm_timeline.py --shape $sh --m 3000 --tile 3,15,19,21 --ksplit 1,2,3 --partials --conv --brief
done
python tools/gemm_timeline.py --shape w13 --m 3000 --tile 15,19,21,29 --conv --brief
python tools/gemm_timeline.py --shape fc2 --m 3480 --tile 3,15,19 --ksplit 1,2 --partials --brief
python tools/gemm_timeline.py --shape qkv --m 3000 --tile 15,19,25,29 --brief
