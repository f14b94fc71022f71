"""elg_amd -- MI355X-native ELG-POMO rollout engine (drop-in for gaocrr/ELG's construction hot path)."""
__version__ = "0.1.0"
