"""PETSc-style option database (``-key value`` on the command line), replacing
``petsc4py.init(sys.argv)`` + ``PETSc.Options()`` (src/run_case.py:4-10,165-167).  Keys are
stored without the leading dash."""
import sys


class Options:
    _db = None

    def __init__(self, argv=None):
        if Options._db is None or argv is not None:
            Options._db = self._parse(sys.argv[1:] if argv is None else argv)

    @staticmethod
    def _parse(args):
        db = {}
        i = 0
        while i < len(args):
            a = args[i]
            if a.startswith('-') and not _is_number(a):
                key = a.lstrip('-')
                if i + 1 < len(args) and (not args[i + 1].startswith('-') or _is_number(args[i + 1])):
                    db[key] = args[i + 1]
                    i += 2
                    continue
                db[key] = True
            i += 1
        return db

    def hasName(self, key):
        return key in Options._db

    def getString(self, key, default=None):
        v = Options._db.get(key, default)
        return default if v is True else v

    def getInt(self, key, default=None):
        v = Options._db.get(key)
        return default if v is None or v is True else int(v)

    def getReal(self, key, default=None):
        v = Options._db.get(key)
        return default if v is None or v is True else float(v)

    def setValue(self, key, value):
        Options._db[key.lstrip('-')] = str(value)

    def delValue(self, key):
        Options._db.pop(key.lstrip('-'), None)


def _is_number(s):
    try:
        float(s)
        return True
    except ValueError:
        return False
