"""Small helpers shared by the host classes."""
import importlib
from typing import Any


def import_and_get(name: str) -> Any:
    """Resolve "package.module.Attr" (config strings `game:` / `network:`, utils.py:6-15)."""
    module_name, sep, attr = name.rpartition(".")
    if not sep:
        raise ImportError("expected <module>.<name>, got %r" % name)
    module = importlib.import_module(module_name)
    try:
        return getattr(module, attr)
    except AttributeError:
        raise ImportError("%s has no attribute %s" % (module_name, attr)) from None
