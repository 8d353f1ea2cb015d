"""Wire packets between load generator, engines and the orchestrator.

Field-for-field the reference's packets (utils/packets.py:6-22, :32-59): they are
pickled through multiprocessing.Queue, so attribute names are the contract.  Note
the reference stores the `process_start_time` argument as `queue_start_time`.
"""


class ServiceRequest(object):
    # model_id is this build's extension for mixed-model streams (which of the engine's models
    # the query is for); engines read it with getattr(..., 0) so reference packets still work
    __slots__ = ("batch_id", "batch_size", "epoch", "arrival_time", "total_sub_batches", "sub_id",
                 "exp_packet", "model_id")

    def __init__(self, batch_id=None, epoch=None, arrival_time=None, batch_size=None, sub_id=None,
                 total_sub_batches=None, exp_packet=None, model_id=0):
        self.model_id = model_id
        self.batch_id = batch_id
        self.batch_size = batch_size
        self.epoch = epoch
        self.arrival_time = arrival_time
        self.total_sub_batches = total_sub_batches
        self.sub_id = sub_id
        self.exp_packet = exp_packet

    def __reduce__(self):
        # (pickled through multiprocessing queues ~1 M times a second at 8 GPUs: a positional tuple instead of the
        #  default per-attribute state -- 0.25 us instead of 2 us per packet; the attributes are unchanged)
        return (ServiceRequest, (self.batch_id, self.epoch, self.arrival_time, self.batch_size, self.sub_id,
                                 self.total_sub_batches, self.exp_packet, self.model_id))

    def __str__(self):
        # the reference's __str__ raises (packets.py:24-27); this one works
        return "Request[%s] -> arrival_time %s" % ((self.epoch, self.batch_id, self.batch_size),
                                                   self.arrival_time)


class ServiceResponse(object):
    __slots__ = ("consumer_id", "epoch", "batch_id", "batch_size", "arrival_time", "queue_start_time",
                 "queue_end_time", "inference_end_time", "out_batch_size", "total_sub_batches",
                 "exp_packet", "sub_id", "model_id")

    def __init__(self, consumer_id=None, epoch=None, batch_id=None, batch_size=None,
                 arrival_time=None, process_start_time=None, queue_end_time=None,
                 inference_end_time=None, out_batch_size=None, sub_id=None, total_sub_batches=None,
                 exp_packet=None, model_id=0):
        self.model_id = model_id
        self.consumer_id = consumer_id
        self.epoch = epoch
        self.batch_id = batch_id
        self.batch_size = batch_size
        self.arrival_time = arrival_time
        self.queue_start_time = process_start_time
        self.queue_end_time = queue_end_time
        self.inference_end_time = inference_end_time
        self.out_batch_size = out_batch_size
        self.total_sub_batches = total_sub_batches
        self.exp_packet = exp_packet
        self.sub_id = sub_id

    def __reduce__(self):
        return (ServiceResponse, (self.consumer_id, self.epoch, self.batch_id, self.batch_size, self.arrival_time,
                                  self.queue_start_time, self.queue_end_time, self.inference_end_time, self.out_batch_size,
                                  self.sub_id, self.total_sub_batches, self.exp_packet, self.model_id))

    def as_dict(self, with_model=False):
        """What the orchestrator logs per response (reference uses response.__dict__): the
        reference's fields; mixed-model runs add model_id."""
        return {k: getattr(self, k) for k in self.__slots__ if with_model or k != "model_id"}

    def __str__(self):
        return "Response[%s] -> arrival %s start %s end %s inference_end %s" % (
            (self.epoch, self.batch_id, self.batch_size, self.consumer_id), self.arrival_time,
            self.queue_start_time, self.queue_end_time, self.inference_end_time)


class ResponseBlock(object):
    """Many whole-query responses of ONE engine in one put (this build, `--accel_response_blocks n`): the fields of the
    reference's ServiceResponse as columns.  What an accelerator engine sends instead of n ServiceResponse objects when
    the orchestrator would otherwise be the bottleneck (eight MI355X answer ~1.5 M queries/s; a Python process
    unpickles and books ~0.5 M packets/s): the orchestrator books a block with a handful of numpy operations and turns
    it back into per-response records only for its log.  Every response in a block is a whole query
    (total_sub_batches 1, sub_id 0), which is all an accelerator engine ever answers."""
    __slots__ = ("consumer_id", "epoch", "batch_id", "batch_size", "arrival_time", "queue_start_time",
                 "inference_end_time", "exp_packet", "model_id")

    def __init__(self, consumer_id, epoch, batch_id, batch_size, arrival_time, queue_start_time, inference_end_time,
                 exp_packet, model_id):
        import numpy as np
        self.consumer_id = consumer_id
        self.epoch = np.asarray(epoch, dtype=np.int32)
        self.batch_id = np.asarray(batch_id, dtype=np.int32)
        self.batch_size = np.asarray(batch_size, dtype=np.int32)
        self.arrival_time = np.asarray(arrival_time, dtype=np.float64)
        self.queue_start_time = np.asarray(queue_start_time, dtype=np.float64)
        self.inference_end_time = np.asarray(inference_end_time, dtype=np.float64)
        self.exp_packet = np.asarray(exp_packet, dtype=np.bool_)
        self.model_id = np.asarray(model_id, dtype=np.int32)

    def __len__(self):
        return int(self.epoch.size)

    def __reduce__(self):
        return (ResponseBlock, (self.consumer_id, self.epoch, self.batch_id, self.batch_size, self.arrival_time,
                                self.queue_start_time, self.inference_end_time, self.exp_packet, self.model_id))

    def responses(self):
        """the block as the ServiceResponse objects it stands for (the orchestrator's log)"""
        for i in range(len(self)):
            end = float(self.inference_end_time[i])
            yield ServiceResponse(consumer_id=self.consumer_id, epoch=int(self.epoch[i]), batch_id=int(self.batch_id[i]),
                                  batch_size=int(self.batch_size[i]), arrival_time=float(self.arrival_time[i]),
                                  process_start_time=float(self.queue_start_time[i]), queue_end_time=end,
                                  inference_end_time=end, out_batch_size=int(self.batch_size[i]), sub_id=0,
                                  total_sub_batches=1, exp_packet=bool(self.exp_packet[i]), model_id=int(self.model_id[i]))
