"""Checkpoint contents beyond the model / optimizer tensors (those are covered in tests/test_trainer*.py)."""


def test_dropout_generator_is_seeded_per_rank_and_its_counter_is_checkpointed(tmp_path):
    """Replicas share ``torch.manual_seed`` but must not draw the same dropout masks; a checkpoint carries the mask
    counter so a resumed run continues the sequence (dropout RNG seed / rank / checkpoint)."""
    import torch
    from distributed_deep_learning_amd import checkpoint as ckpt
    from distributed_deep_learning_amd.ops import kernels as K
    torch.manual_seed(42)
    cpu = torch.device("cpu")
    K.seed_dropout(0)
    s0 = K.rng_state(cpu).tolist()
    K.seed_dropout(1)
    s1 = K.rng_state(cpu).tolist()
    K.seed_dropout(0)
    assert s0[0] != s1[0] and s0[1] == s1[1] == 0 and K.rng_state(cpu).tolist() == s0
    K.rng_state(cpu)[1] += 17                           # 17 masks drawn
    m = torch.nn.Linear(3, 2)
    path = str(tmp_path / "c.pt")
    ckpt.save(path, m, epoch=1, step=5)
    K.seed_dropout(3)                                   # "another rank" resumes: its own seed, the saved counter
    seed3 = K.rng_state(cpu).tolist()[0]
    ckpt.load(path, m)
    assert K.rng_state(cpu).tolist() == [seed3, 17]
    K.seed_dropout(0)
