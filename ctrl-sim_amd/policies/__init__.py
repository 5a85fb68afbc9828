from .policy import Policy  # noqa: F401
from .autoregressive_policy import AutoregressivePolicy  # noqa: F401
