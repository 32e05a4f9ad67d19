from .linear import Linear
