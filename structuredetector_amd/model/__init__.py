from .loss import Loss, LossStats
