"""MI355X-native hot path of ChuhanZhou/News_Recommendation_Model.

Drop-in classes for the reference's ``models`` package; the compute runs in hand-written gfx950 HIP
kernels behind the C ABI of include/nrm_hotpath.h.
"""
from .config import Dims, model_config, WORKLOADS          # noqa: F401
from .modules import (MLP, PointwiseAttention, PointwiseAttentionExpanded,            # noqa: F401
                      UserInstantInterestModel, UserInvariantInterestModel, UserModel)

__all__ = ["MLP", "PointwiseAttention", "PointwiseAttentionExpanded", "UserInstantInterestModel",
           "UserInvariantInterestModel", "UserModel", "Dims", "model_config", "WORKLOADS"]
