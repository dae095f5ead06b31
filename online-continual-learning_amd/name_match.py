"""The reference's plugin registries with the same keys (utils/name_match.py:31-55), restricted to the hot path
BASELINE.json names.  `general_main.py` of the reference becomes a drop-in by importing these dictionaries
instead of its own."""


class _Lazy(dict):
    """Registry whose values are resolved on first access (avoids import cycles: Buffer -> name_match -> plugins)."""

    def __init__(self, table):
        super().__init__()
        self._table = table

    def __missing__(self, key):
        if key not in self._table:
            raise KeyError(key)
        mod, attr = self._table[key]
        import importlib
        val = getattr(importlib.import_module(mod, __package__), attr)
        self[key] = val
        return val

    def __contains__(self, key):
        return key in self._table

    def keys(self):
        return self._table.keys()


agents = _Lazy({
    'ER': ('.agents.exp_replay', 'ExperienceReplay'),
    'SCR': ('.agents.scr', 'SupContrastReplay'),
})

retrieve_methods = _Lazy({
    'MIR': ('.plugins.mir_retrieve', 'MIR_retrieve'),
    'random': ('.plugins.random_retrieve', 'Random_retrieve'),
    'ASER': ('.plugins.aser_retrieve', 'ASER_retrieve'),
    'match': ('.plugins.sc_retrieve', 'Match_retrieve'),
    'mem_match': ('.plugins.mem_match', 'MemMatch_retrieve'),
})

update_methods = _Lazy({
    'random': ('.plugins.reservoir_update', 'Reservoir_update'),
    'GSS': ('.plugins.gss_greedy_update', 'GSSGreedyUpdate'),
    'ASER': ('.plugins.aser_update', 'ASER_update'),
})
