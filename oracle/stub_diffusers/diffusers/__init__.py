"""Test-only stand-in for the absent ``diffusers`` package so that the reference's OWN files
(lora.py, train_util.py, model_util.py, config_util.py) can be imported read-only from
/root/reference by tests and by tests/golden/make_golden.py.  It only supplies the class
NAMES those files import; all arithmetic comes from oracle/unet_ref.py and oracle/ddim_ref.py."""


class UNet2DConditionModel:  # type annotation only in the reference
    pass


class SchedulerMixin:
    pass


class StableDiffusionPipeline:
    pass


class StableDiffusionXLPipeline:
    pass
