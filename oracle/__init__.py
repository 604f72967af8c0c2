"""ORACLE — TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference's query-time scoring path (see oracle/*.h headers for the
reference file:line each function follows). Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this package, and only as the checker / reported CPU baseline.
The product (typesense_amd/, libtsgpu.so) never imports, links or calls anything in here.
"""
