cd $GRAFT_REPO_ROOT
for a in 0 1 2 3 4 7 8 24 31; do COFI_ATTN_ABLATE=$a python tools/attn_ablate.py 1280 1280 1 2>&1 | tail -1; done
COFI_ATTN_ABLATE=0 python tools/attn_ablate.py 1280 1280 2 2>&1 | tail -1
COFI_ATTN_ABLATE=0 python tools/attn_ablate.py 1280 1280 16 2>&1 | tail -1
