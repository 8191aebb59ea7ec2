from .loss import Loss, LossStats
from .network import Network
