"""Batched mirror of the reference's `ValueSimLP` (agents/ValueSimLP.py:7-75): expand first, evaluate the
leaf's <=7 unique child observations in one batch, averaged leaf-parallel backup (core.h:303-381)."""
from .. import store as st
from .ValueSim import ValueSim


class ValueSimLP(ValueSim):
    kind = st.KIND_VALUESIM_LP

    def __init__(self, **kwargs):
        kwargs.setdefault("min_visits_to_store", 25)
        super().__init__(**kwargs)
