"""flow-matching transport (dimsum/transport/__init__.py:5-69)"""
from .transport import ModelType, PathType, Sampler, Transport, WeightType  # noqa: F401


def create_transport(path_type="Linear", prediction="velocity", loss_weight=None, train_eps=None, sample_eps=None,
                     path_args={}, t_sample_mode="uniform"):
    model_type = {"noise": ModelType.NOISE, "score": ModelType.SCORE}.get(prediction, ModelType.VELOCITY)
    loss_type = {"velocity": WeightType.VELOCITY, "likelihood": WeightType.LIKELIHOOD}.get(loss_weight, WeightType.NONE)
    ptype = {"Linear": PathType.LINEAR, "GVP": PathType.GVP, "VP": PathType.VP}[path_type]
    if ptype is PathType.VP:
        train_eps = 1e-5 if train_eps is None else train_eps
        sample_eps = 1e-3 if train_eps is None else sample_eps
    elif model_type is not ModelType.VELOCITY:
        train_eps = 1e-3 if train_eps is None else train_eps
        sample_eps = 1e-3 if train_eps is None else sample_eps
    else:       # velocity prediction on GVP / Linear paths is stable on the whole interval
        train_eps = sample_eps = 0
    return Transport(model_type=model_type, path_type=ptype, loss_type=loss_type, train_eps=train_eps, sample_eps=sample_eps,
                     path_args=path_args, t_sample_mode=t_sample_mode)
