"""placeholder -- filled in later this round"""
