"""`from models.<package> import *` exposes every model module of this package by name (the reference's
main.py resolves model classes that way); main.py here imports the module it needs directly."""
import pkgutil

__all__ = [m.name for m in pkgutil.iter_modules(__path__)]
