"""ORACLE — TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference's hot path; imported only by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg.  The product package
`ganecdotes_amd` never imports it.
"""
