# model modules are discovered by name, as in the reference (models/general/__init__.py:1-7)
from os.path import basename, dirname, isfile, join
import glob

__all__ = [basename(f)[:-3] for f in glob.glob(join(dirname(__file__), "*.py"))
           if isfile(f) and not f.endswith('__init__.py')]
