"""
sugar_amd -- MI355X-native embedding-extraction path of EfficientTDNN (mechanicalsea/sugar).

Python keeps sugar's call surface (WrappedModel.from_pretrained / subnet(wav) / extract_vectors / score_trials)
and its checkpoint format; all device work is hand-written gfx950 HIP behind the C-ABI of include/sugar_amd.h.
"""
from .dynamictdnn import (DynamicTDNN, TDNNPath, TDNNSubnet, ecapa_tdnn, specfic_tdnn, tdnn8m2g, tdnn14m4g,
                          tdnn16m4g, tdnn26m7g)
from .graphs import GraphedEmbedder
from .latency import TDNNLatency, latency
from .metrics import calculate_eer, calculate_mindcf, calculate_val
from .models import (SpeakerModel, WrappedModel, batch_forward, bn_state_dict, checkpoint2backbone, keep_params,
                     load_bn_state_dict, veri_validate)
from .scores import (asnorm, calculate_score, load_sctr, normalize_scores, save_sctr, score_batch, score_cohorts,
                     score_trials, snorm, ztnorm)
from .transforms import LogMelFbanks, PreEmphasis
from .nas import AccuracyPredictor, ArchCoder, GPULatencyConstraint, MinimizeError, config2d, ea_solve
from .sweep import evaluate_subnets, gather_results, shard_configs
from .vectors import PackedUtterances, VectorTable, extract_vectors, length_normalize, vector_transform

__all__ = ["DynamicTDNN", "TDNNPath", "TDNNSubnet", "ecapa_tdnn", "specfic_tdnn", "tdnn8m2g", "tdnn14m4g",
           "tdnn16m4g", "tdnn26m7g", "SpeakerModel", "WrappedModel", "bn_state_dict", "checkpoint2backbone",
           "calculate_score", "score_trials", "LogMelFbanks", "PreEmphasis", "extract_vectors", "veri_validate", "batch_forward",
           "calculate_eer", "calculate_mindcf", "score_cohorts", "asnorm", "snorm", "TDNNLatency", "latency", "evaluate_subnets",
           "shard_configs", "gather_results", "PackedUtterances", "VectorTable",
           "calculate_val", "load_bn_state_dict", "keep_params", "ztnorm", "score_batch", "normalize_scores", "save_sctr", "load_sctr", "length_normalize", "vector_transform",
           "ArchCoder", "config2d", "AccuracyPredictor", "MinimizeError", "GPULatencyConstraint", "ea_solve", "GraphedEmbedder"]
