"""
Results that stay on the device until a script asks for them.

The reference fills the accountants of every surface and the whole ray tree on the host while it traces
(tracer/optics_callables.py:1577-1643, :1949-1961; tracer/trace_tree.py:6-55, read back through
tracer/tracer_engine.py:288-295).  Here a trace leaves them where the kernels wrote them -- the hit buffer of the fast
engine, the levels of the ordered engine -- and hands every accountant concerned a *mark*: a promise of data in its
list.  The first `get_data()` / `get_all_hits()` that meets a mark settles the delivery it stands for: the data are
copied (page-locked memory, one copy per column), fed to every accountant that still holds a mark of it, and the marks
disappear.  An accountant that was reset in between has dropped its mark and gets nothing, as if it had been fed and
then reset; a delivery nobody holds a mark of any more is never fetched.

Deliveries are settled oldest first, whichever accountant asks, so that the chunks of all accountants stay in the order
of the calls (the columns of a surface's accountants -- energies, hit points, directions -- must line up ray by ray).
"""
import weakref

_UNSETTLED = []          # weak references to the deliveries not settled yet, oldest first


class Mark(object):
    """placeholder in an accountant's list of chunks: the data of `delivery` for this accountant, not fetched yet"""
    __slots__ = ('delivery', '__weakref__')

    def __init__(self, delivery):
        self.delivery = delivery


class Delivery(object):
    """what one trace (or, for the hit buffer of the fast engine, several) owes to the accountants"""
    def __init__(self):
        self.settled = False
        self._marks = weakref.WeakSet()
        self._accs = []                 # weak references to the accountants that were given a mark
        self.always = False             # deliver even when no accountant holds a mark (side results: the transfer matrix)
        _UNSETTLED.append(weakref.ref(self))

    # -- marks -----------------------------------------------------------------------------------
    def holds_mark(self, acc):
        return any(isinstance(c, Mark) and c.delivery is self for c in acc._data)

    def give_mark(self, acc):
        if self.holds_mark(acc):
            return
        m = Mark(self)
        acc._data.append(m)
        self._marks.add(m)
        self._accs.append(weakref.ref(acc))

    def wanted(self):
        """some accountant still waits for this delivery"""
        return not self.settled and (self.always or len(self._marks) > 0)

    # -- settling ----------------------------------------------------------------------------------
    def settle(self):
        """deliver this and every older delivery (oldest first)"""
        if self.settled:
            return
        for ref in list(_UNSETTLED):
            d = ref()
            if d is None or d.settled:
                continue
            if d is self:
                break
            d._settle_one()
        self._settle_one()

    def _settle_one(self):
        if self.settled:
            return
        self.settled = True
        holders = set()
        for ref in self._accs:
            acc = ref()
            if acc is None:
                continue
            kept = [c for c in acc._data if not (isinstance(c, Mark) and c.delivery is self)]
            if len(kept) != len(acc._data):
                holders.add(id(acc))
                acc._data = kept
        self._accs = []
        try:
            if holders or self.always:
                self.deliver(holders)
        finally:
            self.release()
            _UNSETTLED[:] = [r for r in _UNSETTLED if r() is not None and not r().settled]

    # -- for subclasses ------------------------------------------------------------------------------
    def deliver(self, holders):
        """fetch the data and feed the accountants whose id() is in `holders`"""
        raise NotImplementedError

    def release(self):
        """the device side is no longer needed"""


def settle_marks(acc):
    """called by an accountant before it reads its chunks: afterwards its list holds arrays only"""
    while True:
        m = next((c for c in acc._data if isinstance(c, Mark)), None)
        if m is None:
            return
        m.delivery.settle()


def settle_all():
    for ref in list(_UNSETTLED):
        d = ref()
        if d is not None:
            d.settle()
