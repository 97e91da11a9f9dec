python -m pytest tests/test_model_ops_parity.py -x -q -m gpu -k "gemm" 2>&1 | tail -3
for sh in 25600,768,2304 25600,768,3072 25600,3072,768 6400,1536,4608 102400,384,1152 409600,192,576; do python tools/gemm_shapes.py --shape $sh --no-library 2>&1 | grep shape; done
