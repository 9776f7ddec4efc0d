import sys, os
sys.path.insert(0, os.getcwd())
import acg_alp_ldpc_amd as A
from acg_alp_ldpc_amd import _lib
H = A.read_pcm("data/H05.txt")
G, _ = H.get_orthogonal()
cws = A.gen_random_codewords(G, 8192, 239239239)
for prec, tag in ((_lib.PREC_F64, "fp64"), (_lib.PREC_F32, "fp32")):
    dec = A.QPADMMDecoder(1.95, 0.5, 100, 1e-5, precision=prec)
    for snr in (-2.0, 0.0, 2.0):
        A.run_experiment(dec, cws, H, snr, frames=4096, noise="device", seed=1)
        r = A.run_experiment(dec, cws, H, snr, frames=1 << 19, noise="device", seed=1)
        print("QP-ADMM-100 %s %+.1f dB: %6.2f M frames/s FER %.5f mean sweeps %.2f layout %s" % (tag, snr, (1 << 19) / (r.kernel_ms * 1e-3) / 1e6, r.FER(), r.mean_iters(), dec.layout(H)))
    dec.close()
