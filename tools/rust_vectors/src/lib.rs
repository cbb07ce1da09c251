// intentionally empty: see tests/emit.rs
