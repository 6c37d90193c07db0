"""helpers of bench.py (measurement harness, not part of the product)"""
