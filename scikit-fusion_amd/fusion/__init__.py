"""Public fusion API (mirror of reference ``skfusion/fusion/__init__.py:1-2``)."""
from .base import FusionBase, FusionFit, FusionTransform, DataFusionError, save_fit, load_fit
from .fusion_graph import FusionGraph, Relation, ObjectType
from .decomposition import Dfmf, DfmfTransform, Dfmc

__all__ = ['FusionBase', 'FusionFit', 'FusionTransform', 'DataFusionError',
           'FusionGraph', 'Relation', 'ObjectType', 'Dfmf', 'DfmfTransform', 'Dfmc', 'save_fit', 'load_fit']
