"""Optional event log for the parity tests: when LOG is a list, plugins / agents append (tag, dict) records of the
indices and scores they used.  Off (None) in normal operation: no extra device synchronisation."""
LOG = None


def emit(tag, **kw):
    if LOG is not None:
        LOG.append((tag, kw))


def on():
    return LOG is not None
