from .collector import NativeCollector, VecCollector

__all__ = ["VecCollector", "NativeCollector"]
