"""apertis_llm_amd — MI355X (gfx950) implementation of the Apertis-LLM data-parallel hot path.

Public names mirror the reference's `src.model.core` / `src.multimodal.module`; the data formats and the trainer
(`src.training.pipeline`) are `apertis_llm_amd.data` and `apertis_llm_amd.trainer`."""
from ._lib import ApertisHipError  # noqa: F401
from .model import (AdaptiveExpertSystem, ApertisAttention, ApertisConfig, ApertisFeedForward,  # noqa: F401
                    ApertisForCausalLM, ApertisLayer, ApertisModel, RMSNorm, RotaryEmbedding,
                    SelectiveLinearAttention, SwiGLUFFN, calculate_model_dimensions, create_apertis_model,
                    estimate_model_parameters, parse_param_count)
from .multimodal import UnifiedMultimodalEncoder  # noqa: F401

__all__ = ["ApertisConfig", "ApertisModel", "ApertisForCausalLM", "create_apertis_model",
           "estimate_model_parameters", "calculate_model_dimensions", "parse_param_count",
           "UnifiedMultimodalEncoder", "SelectiveLinearAttention", "AdaptiveExpertSystem", "ApertisHipError"]
