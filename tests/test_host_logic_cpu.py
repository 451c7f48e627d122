"""host-side logic that needs no GPU: the merged table of partially frozen embeddings, the fork node's gradient plumbing"""
import copy

import torch

from pasero_amd.modules import Embedding


def test_partially_frozen_embedding_merges_its_tables_once_per_parameter_state():
    """Embedding.effective_weight (pasero/models/modules.py:929-946 as ONE table): the same tensor for every reader until a
    table changes; the merge node's backward routes each row's gradient to the table it came from, also when the node serves
    two graphs; under no_grad nothing is recorded; the module still deep-copies"""
    torch.manual_seed(0)
    mask = torch.tensor([True, False, True, False, False])
    e = Embedding(5, 4, 1, freeze_mask=mask)
    E1, E2 = e.effective_weight(), e.effective_weight()
    assert E1 is E2
    ref = torch.where(mask[:, None], e.frozen_embedding.weight, e.weight)
    assert torch.equal(E1, ref)
    (E1.sum() * 2 + (E2 ** 2).sum()).backward()
    gw, gf = e.weight.grad.clone(), e.frozen_embedding.weight.grad.clone()
    assert (gw[mask] == 0).all() and (gf[~mask] == 0).all() and gf[mask].abs().min() > 0
    e.zero_grad()
    (ref.sum() * 2 + (ref ** 2).sum()).backward()
    assert torch.equal(gw, e.weight.grad) and torch.equal(gf, e.frozen_embedding.weight.grad)
    E3 = e.effective_weight()
    assert E3 is E1
    E3.sum().backward()  # a second graph through the same node
    with torch.no_grad():
        e.weight.add_(1.0)  # what an optimizer step does
    E4 = e.effective_weight()
    assert E4 is not E1 and torch.equal(E4, torch.where(mask[:, None], e.frozen_embedding.weight, e.weight))
    with torch.no_grad():
        assert e.effective_weight().grad_fn is None
    assert e.effective_weight().grad_fn is not None
    copy.deepcopy(e)
    plain = Embedding(5, 4, 1)
    assert plain.effective_weight() is plain.weight


def test_drop_link_refuses_a_gradient_that_autograd_summed_in_place():
    """a second consumer of z, created BEFORE the fork, makes the engine add its gradient to the fork's
    dz — in place when the input buffer holds the last reference, so the sum sits at dz's address.  The link keeps dz alive
    until the hand-over and compares the version counter: the producer must then see `take() is None` and draw its own mask.
    Toy nodes on CPU with the same offer / take protocol as LayerNormForkFn / ResidualDropoutFn."""
    from pasero_amd.autograd import DropLink
    seen = {}

    class Producer(torch.autograd.Function):  # z = residual + dropout(x), mask = every other element
        @staticmethod
        def forward(ctx, x, link):
            ctx.link = link
            return x.clone()

        @staticmethod
        def backward(ctx, dz):
            seen['dz'] = dz.clone()
            seen['masked'] = ctx.link.take(dz)
            return dz, None

    class Fork(torch.autograd.Function):
        @staticmethod
        def forward(ctx, z, link):
            ctx.link = link
            return z * 2.0

        @staticmethod
        def backward(ctx, dy):
            dz = dy * 2.0
            masked = dz.clone()
            masked[::2] = 0
            seen['fork_dz'] = dz.clone()
            ctx.link.offer(masked, dz)
            return dz, None

    # the stock flow: one consumer -> the offer is taken
    x = torch.arange(8.0, requires_grad=True)
    link = DropLink()
    Fork.apply(Producer.apply(x, link), link).sum().backward()
    assert seen['masked'] is not None and torch.equal(seen['dz'], seen['fork_dz'])
    assert link.masked is None and link.dz is None
    # a second consumer created before the fork: the producer receives the SUM and must not use the fork's masked copy
    x = torch.arange(8.0, requires_grad=True)
    link = DropLink()
    z = Producer.apply(x, link)
    other = z * 3.0
    (Fork.apply(z, link).sum() + other.sum()).backward()
    assert torch.equal(seen['dz'], seen['fork_dz'] + 3.0)  # what arrived is the sum ...
    assert seen['masked'] is None  # ... so the offer was refused
    assert torch.equal(x.grad, torch.full((8,), 5.0))
    # an offer that nobody took does not leak into the next pass
    link.offer(torch.zeros(8), torch.zeros(8))
    assert link.take(torch.zeros(8)) is None and link.masked is None
