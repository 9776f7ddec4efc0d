cd "${GRAFT_REPO_ROOT:-/root/repo}"
for v in ${VARIANTS:-main}; do
  if [ "$v" = main ]; then unset ACG_LDPC_LIB; else export ACG_LDPC_LIB=$PWD/acg_alp_ldpc_amd/lib/variants/libacg_$v.so; fi
  timeout -k 10 120 python tools/rate.py --algo bp --engine streamed --frames 1048576 --steps 3 --tag $v | tail -1 || exit 1
  timeout -k 10 120 python tools/rate.py --algo minsum --engine streamed --frames 1048576 --steps 3 --tag $v | tail -1 || exit 1
  timeout -k 10 120 python tools/rate.py --algo minsum --engine streamed --synthetic 5000 10000 3 6 --frames 32768 --snr 2 --steps 2 --tag ${v}_c5 | tail -1 || exit 1
done
