"""placeholder - replaced by the HIP-backed ops below in this commit series."""


def __getattr__(name):
    def _missing(*a, **k):
        raise NotImplementedError(f"ops.{name} not built yet")
    return _missing
