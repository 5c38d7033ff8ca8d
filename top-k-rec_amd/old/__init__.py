"""Legacy entry points of the reference's ``old/`` tree that sit on the hot path."""
