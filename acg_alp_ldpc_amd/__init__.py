"""acg_alp_ldpc_amd — MI355X-native batched LDPC decoding behind the reference's Decoder API.

Only the hot path of GreatDrake/acg-alp-ldpc is here: BP (algo/bp.h), QP-ADMM
(algo/qp_admm.h), the AWGN channel (utils/channel.h) and the Monte-Carlo loop (experiment.h),
implemented as hand-written HIP kernels for gfx950 in csrc/ behind the C ABI of include/acg_ldpc.h.
"""
from ._lib import (ENGINE_AUTO, ENGINE_FUSED, ENGINE_STREAMED, PREC_DEFAULT, PREC_F16, PREC_F32, PREC_F64,  # noqa: F401
                   SCHEDULE_FLOODING, SCHEDULE_LAYERED, LdpcError, build, lib)
from .channel import gen_random_codewords, llr, llr_variance, transmit_frames  # noqa: F401
from .code import ParityCheckMatrix  # noqa: F401
from .codes import regular_ldpc  # noqa: F401
from .decoder import BeliefPropagationDecoder, Decoder, MinSumDecoder, QPADMMDecoder  # noqa: F401
from .experiment import (ExperimentResult, merge_exp_results, run_experiment,  # noqa: F401
                         run_experiment_inproc, run_experiment_sharded, shard_range)


def read_pcm(path):
    """utils/parse_data.h:6-25 -> ParityCheckMatrix"""
    return ParityCheckMatrix.read_pcm(path)


def device_available():
    return bool(lib().acg_ldpc_device_available())
