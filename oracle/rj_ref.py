"""Oracle for the reversible-jump move bookkeeping (reference: sampler/reversible_jump.py).
TEST INFRASTRUCTURE ONLY.  Integer / index path: results must match bit for bit."""


def move_type(n, n_max, birth_probability, u):
    """(birth, consumed_uniform)  [reversible_jump.py:310-333]: at n == n_max a death, at n == 1 a
    birth (no uniform is drawn in either case), n == 0 is an error, else birth iff u <= q."""
    if n == n_max:
        return False, False
    if n == 1:
        return True, False
    if n == 0:
        raise ValueError("Reversible jump MCMC: Number of parameters cannot be zero.")
    return bool(u <= birth_probability), True


def move_probabilities(n, n_max, birth_probability, birth):
    """(p_birth, p_death) with the edge cases at n_max, n_max-1, 1 and 2  [reversible_jump.py:335-373]."""
    p_birth, p_death = birth_probability, 1.0 - birth_probability
    if n == n_max:
        p_death = 1.0
    if n == n_max - 1 and birth:
        p_death = 1.0
    if n == 1:
        p_birth = 1.0
    if n == 2 and not birth:
        p_birth = 1.0
    return p_birth, p_death
