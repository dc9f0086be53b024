class DDIMScheduler:
    pass


class DDPMScheduler:
    pass


class LMSDiscreteScheduler:
    pass


class EulerAncestralDiscreteScheduler:
    pass
