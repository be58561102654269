"""placeholder, replaced below"""
