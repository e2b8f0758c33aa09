from .dfmf import Dfmf, DfmfTransform
from .dfmc import Dfmc

__all__ = ['Dfmf', 'DfmfTransform', 'Dfmc']
