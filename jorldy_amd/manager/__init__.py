from .collector import VecCollector

__all__ = ["VecCollector"]
