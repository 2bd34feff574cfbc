"""Parts of bench.py (the entry point at the repository root): common constants, the call-protocol back ends, the lock-stepped legs, the
oracle-side checkers and the report writers."""
